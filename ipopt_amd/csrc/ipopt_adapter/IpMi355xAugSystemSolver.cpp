// IpMi355xAugSystemSolver.cpp -- see the header.
#include "IpMi355xAugSystemSolver.hpp"
#include "IpMi355xSolverInterface.hpp"
#include "IpTripletHelper.hpp"
#include "IpIpoptData.hpp"
#include "IpTimingStatistics.hpp"
#include "IpSymTMatrix.hpp"
#include "IpGenTMatrix.hpp"
#include "IpDenseVector.hpp"
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace Ipopt
{

// Leaf copies into the pinned staging buffers.  TripletHelper::FillValues (IpTripletHelper.cpp:249-362) ends, for the matrix types
// a TNLP produces, in one single-threaded copy of Values(); at n = 10^6 those copies (56 MB per iteration) are a third of what is
// left of PDSystemSolverTotal once everything else runs on the GPU, so the plain types are copied by a few threads here and
// everything else goes through TripletHelper as before.
static void ParallelCopy(Number* dst, const Number* src, size_t n)
{
   const size_t chunk = (size_t) 1 << 19;        // doubles per thread at least (4 MB)
   const int nt = (int) std::min<size_t>(8, n / chunk);
   if( nt <= 1 )
   {
      std::memcpy(dst, src, n * sizeof(Number));
      return;
   }
   std::vector<std::thread> th;
   const size_t per = (n + nt - 1) / nt;
   for( int t = 0; t < nt; ++t )
   {
      const size_t o = (size_t) t * per;
      if( o >= n )
      {
         break;
      }
      const size_t len = std::min(per, n - o);
      th.emplace_back([=]() { std::memcpy(dst + o, src + o, len * sizeof(Number)); });
   }
   for( size_t t = 0; t < th.size(); ++t )
   {
      th[t].join();
   }
}

static void FillMatrixValues(Index n_entries, const Matrix& M, Number* dst)
{
   if( const SymTMatrix* st = dynamic_cast<const SymTMatrix*>(&M) )
   {
      ParallelCopy(dst, st->Values(), (size_t) n_entries);
   }
   else if( const GenTMatrix* gt = dynamic_cast<const GenTMatrix*>(&M) )
   {
      ParallelCopy(dst, gt->Values(), (size_t) n_entries);
   }
   else
   {
      TripletHelper::FillValues(n_entries, M, dst);
   }
}

static void FillVectorValues(Index n, const Vector& v, Number* dst)
{
   const DenseVector* d = dynamic_cast<const DenseVector*>(&v);
   if( d && !d->IsHomogeneous() )
   {
      ParallelCopy(dst, d->Values(), (size_t) n);
   }
   else
   {
      TripletHelper::FillValuesFromVector(n, v, dst);
   }
}

Mi355xAugSystemSolver::Mi355xAugSystemSolver()
   : handle_(NULL), structured_(false), analysed_(false), have_factor_(false), pivtol_changed_(false),
     warm_start_same_structure_(false), pivtol_(1e-8), pivtolmax_(1e-4), negevals_(-1), n_x_(0), n_s_(0), n_c_(0), n_d_(0),
     dim_(0), nnz_(0), nnz_w_(0), nnz_jc_(0), nnz_jd_(0), w_tag_(0), jc_tag_(0), jd_tag_(0), dx_tag_(0), ds_tag_(0), dc_tag_(0),
     dd_tag_(0), uploaded_bytes_(0), nfact_noupload_(0), pd_wanted_(false), pd_defined_(false), singular_(false)
{
   mi355x_kkt_default_options(&kopts_);
   for( int q = 0; q < NSEG; ++q )
   {
      seg_off_[q] = seg_len_[q] = 0;
      scale_[q] = shift_[q] = 0.;
   }
}

Mi355xAugSystemSolver::~Mi355xAugSystemSolver()
{
   if( handle_ )
   {
      mi355x_kkt_destroy(handle_);
   }
}

bool Mi355xAugSystemSolver::InitializeImpl(const OptionsList& options, const std::string& prefix)
{
   Mi355xSolverInterface::ReadNumericOptions(options, prefix, kopts_, pivtol_, pivtolmax_);
   // several Ipopt processes may share the factorisation on this route too: device assembly and the 8-block kernels are replicated work on
   // identical data, the factorisation / solves underneath are the distributed ones (IpMi355xCommBootstrap.hpp)
   comm_.ReadOptions(options, prefix, kopts_);
   bool ws = false;
   try
   {
      options.GetBoolValue("warm_start_same_structure", ws, prefix);
   }
   catch( ... )
   { }
   warm_start_same_structure_ = ws;
   if( !warm_start_same_structure_ || !handle_ )
   {
      if( handle_ )
      {
         mi355x_kkt_destroy(handle_);
         handle_ = NULL;
      }
      if( mi355x_kkt_create(&handle_, &kopts_) != MI355X_KKT_SUCCESS )
      {
         return false;
      }
      structured_ = false;
      analysed_ = false;
      pd_defined_ = false;
      // a new handle starts without a primal-dual workspace REQUEST as well: the dimensions and bound positions of the previous NLP
      // (ReOptimizeTNLP with changed bounds, an AlgorithmBuilder reused for another problem) must not define the new workspace
      pd_wanted_ = false;
      for( int q = 0; q < 4; ++q )
      {
         pd_idx_[q].clear();
      }
   }
   else
   {
      ASSERT_EXCEPTION(structured_, INVALID_WARMSTART,
                       "Mi355xAugSystemSolver called with warm_start_same_structure, but the augmented system is not initialized.");
      mi355x_kkt_set_pivtol(handle_, pivtol_);
      mi355x_kkt_set_pivtolmax(handle_, pivtolmax_);
   }
   have_factor_ = false;
   singular_ = false;
   pivtol_changed_ = false;
   negevals_ = -1;
   w_tag_ = jc_tag_ = jd_tag_ = dx_tag_ = ds_tag_ = dc_tag_ = dd_tag_ = 0;
   return true;
}

void Mi355xAugSystemSolver::BuildStructure(const SymMatrix& W, const Matrix& J_c, const Matrix& J_d)
{
   // block order and offsets of IpStdAugSystemSolver.cpp:263-298; entry order of TripletHelper::FillRowCol_(CompoundSymMatrix)
   // (IpTripletHelper.cpp:805-842): block rows top to bottom, inside a block row the columns left to right
   n_x_ = J_c.NCols();
   n_s_ = J_d.NRows();
   n_c_ = J_c.NRows();
   n_d_ = n_s_;
   dim_ = n_x_ + n_s_ + n_c_ + n_d_;
   nnz_w_ = TripletHelper::GetNumberEntries(W);
   nnz_jc_ = TripletHelper::GetNumberEntries(J_c);
   nnz_jd_ = TripletHelper::GetNumberEntries(J_d);
   const long long len[NSEG] = {nnz_w_, n_x_, n_s_, nnz_jc_, n_c_, nnz_jd_, n_s_, n_d_};
   long long off = 0;
   for( int q = 0; q < NSEG; ++q )
   {
      seg_off_[q] = off;
      seg_len_[q] = len[q];
      off += len[q];
   }
   nnz_ = (Index) off;
   irn_.assign(nnz_ > 0 ? nnz_ : 1, 0);
   jcn_.assign(nnz_ > 0 ? nnz_ : 1, 0);
   // (1,1): W, then the diagonal D_x + delta_x I as SEPARATE (duplicate) entries, exactly like SumSymMatrix does
   TripletHelper::FillRowCol(nnz_w_, W, &irn_[seg_off_[SEG_W]], &jcn_[seg_off_[SEG_W]], 0, 0);
   for( Index i = 0; i < n_x_; ++i )
   {
      irn_[seg_off_[SEG_DX] + i] = jcn_[seg_off_[SEG_DX] + i] = i + 1;
   }
   for( Index i = 0; i < n_s_; ++i )
   {
      irn_[seg_off_[SEG_DS] + i] = jcn_[seg_off_[SEG_DS] + i] = n_x_ + i + 1;
   }
   if( nnz_jc_ > 0 )
   {
      TripletHelper::FillRowCol(nnz_jc_, J_c, &irn_[seg_off_[SEG_JC]], &jcn_[seg_off_[SEG_JC]], n_x_ + n_s_, 0);
   }
   for( Index i = 0; i < n_c_; ++i )
   {
      irn_[seg_off_[SEG_DC] + i] = jcn_[seg_off_[SEG_DC] + i] = n_x_ + n_s_ + i + 1;
   }
   if( nnz_jd_ > 0 )
   {
      TripletHelper::FillRowCol(nnz_jd_, J_d, &irn_[seg_off_[SEG_JD]], &jcn_[seg_off_[SEG_JD]], n_x_ + n_s_ + n_c_, 0);
   }
   for( Index i = 0; i < n_s_; ++i )
   {
      irn_[seg_off_[SEG_ID] + i] = n_x_ + n_s_ + n_c_ + i + 1;   // (4,2): -I
      jcn_[seg_off_[SEG_ID] + i] = n_x_ + i + 1;
   }
   for( Index i = 0; i < n_d_; ++i )
   {
      irn_[seg_off_[SEG_DD] + i] = jcn_[seg_off_[SEG_DD] + i] = n_x_ + n_s_ + n_c_ + i + 1;
   }
   first_vals_.assign(nnz_ > 0 ? nnz_ : 1, 0.);
   structured_ = true;
}

bool Mi355xAugSystemSolver::UpdateSources(const SymMatrix* W, Number W_factor, const Vector* D_x, Number delta_x,
      const Vector* D_s, Number delta_s, const Matrix& J_c, const Vector* D_c, Number delta_c, const Matrix& J_d,
      const Vector* D_d, Number delta_d, bool upload)
{
   // the same change test as StdAugSystemSolver::AugmentedSystemRequiresChange (IpStdAugSystemSolver.cpp:468-540), but
   // piecewise: a source is re-filled (TripletHelper leaf copies) and re-uploaded only when ITS tag changed
   bool changed = !have_factor_;
   double sc[NSEG], sh[NSEG];
   sc[SEG_W] = W ? W_factor : 0.;
   sh[SEG_W] = 0.;
   sc[SEG_DX] = D_x ? 1. : 0.;
   sh[SEG_DX] = delta_x;
   sc[SEG_DS] = D_s ? 1. : 0.;
   sh[SEG_DS] = delta_s;
   sc[SEG_JC] = 1.;
   sh[SEG_JC] = 0.;
   sc[SEG_DC] = D_c ? 1. : 0.;
   sh[SEG_DC] = -delta_c;
   sc[SEG_JD] = 1.;
   sh[SEG_JD] = 0.;
   sc[SEG_ID] = 0.;
   sh[SEG_ID] = -1.;
   sc[SEG_DD] = D_d ? 1. : 0.;
   sh[SEG_DD] = -delta_d;
   for( int q = 0; q < NSEG; ++q )
   {
      if( sc[q] != scale_[q] || sh[q] != shift_[q] )
      {
         changed = true;
      }
      scale_[q] = sc[q];
      shift_[q] = sh[q];
   }
   struct Src
   {
      int seg;
      const TaggedObject* obj;
      TaggedObject::Tag* tag;
   };
   const Src srcs[7] = {{SEG_W, W, &w_tag_}, {SEG_DX, D_x, &dx_tag_}, {SEG_DS, D_s, &ds_tag_}, {SEG_JC, &J_c, &jc_tag_},
      {SEG_DC, D_c, &dc_tag_}, {SEG_JD, &J_d, &jd_tag_}, {SEG_DD, D_d, &dd_tag_}
   };
   for( int k = 0; k < 7; ++k )
   {
      const int q = srcs[k].seg;
      if( !srcs[k].obj || seg_len_[q] == 0 )
      {
         continue;    // scale 0: the source is not read
      }
      if( srcs[k].obj->GetTag() == *srcs[k].tag )
      {
         continue;
      }
      changed = true;
      Number* dst = upload ? mi355x_kkt_assembly_buffer(handle_, q) : &first_vals_[seg_off_[q]];
      switch( q )
      {
         case SEG_W:
            FillMatrixValues(nnz_w_, *W, dst);
            break;
         case SEG_JC:
            FillMatrixValues(nnz_jc_, J_c, dst);
            break;
         case SEG_JD:
            FillMatrixValues(nnz_jd_, J_d, dst);
            break;
         case SEG_DX:
            FillVectorValues(n_x_, *D_x, dst);
            break;
         case SEG_DS:
            FillVectorValues(n_s_, *D_s, dst);
            break;
         case SEG_DC:
            FillVectorValues(n_c_, *D_c, dst);
            break;
         default:
            FillVectorValues(n_d_, *D_d, dst);
            break;
      }
      if( upload )
      {
         mi355x_kkt_assembly_upload(handle_, q);
         uploaded_bytes_ += 8ll * seg_len_[q];
      }
      *srcs[k].tag = srcs[k].obj->GetTag();
   }
   return changed;
}

ESymSolverStatus Mi355xAugSystemSolver::EnsureFactorization(const SymMatrix* W, Number W_factor, const Vector* D_x, Number delta_x,
      const Vector* D_s, Number delta_s, const Matrix* J_c, const Vector* D_c, Number delta_c, const Matrix* J_d,
      const Vector* D_d, Number delta_d, bool check_NegEVals, Index numberOfNegEVals)
{
   ESymSolverStatus retval = SYMSOLVER_SUCCESS;
   do
   {
      if( !structured_ )
      {
         if( !W )
         {
            retval = SYMSOLVER_FATAL_ERROR;   // W must exist during the first call to set up the structure
            break;
         }
         BuildStructure(*W, *J_c, *J_d);
      }
      bool new_matrix;
      if( !analysed_ )
      {
         // lazy analysis with the first values (2x2 pre-pairing wants them), assembled ONCE on the host
         UpdateSources(W, W_factor, D_x, delta_x, D_s, delta_s, *J_c, D_c, delta_c, *J_d, D_d, delta_d, false);
         std::vector<Number> vals(nnz_ > 0 ? nnz_ : 1);
         for( int q = 0; q < NSEG; ++q )
         {
            for( long long i = 0; i < seg_len_[q]; ++i )
            {
               vals[seg_off_[q] + i] = (scale_[q] != 0. ? scale_[q] * first_vals_[seg_off_[q] + i] : 0.) + shift_[q];
            }
         }
         if( HaveIpData() )
         {
            IpData().TimingStats().LinearSystemSymbolicFactorization().Start();
         }
         int st = mi355x_kkt_analyse(handle_, dim_, nnz_, &irn_[0], &jcn_[0], MI355X_KKT_FMT_TRIPLET, &vals[0]);
         if( st == MI355X_KKT_SUCCESS && comm_.Wanted(kopts_) && !comm_.Ready() && !comm_.Setup(handle_, kopts_, Jnlst()) )
         {
            st = MI355X_KKT_FATAL;
         }
         if( st == MI355X_KKT_SUCCESS )
         {
            st = mi355x_kkt_assembly_define(handle_, NSEG, (const int64_t*) seg_off_, (const int64_t*) seg_len_);
         }
         if( st == MI355X_KKT_SUCCESS && pd_wanted_ && !DefinePrimalDualWorkspace() )
         {
            st = MI355X_KKT_FATAL;
         }
         if( HaveIpData() )
         {
            IpData().TimingStats().LinearSystemSymbolicFactorization().End();
         }
         if( st != MI355X_KKT_SUCCESS )
         {
            Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x analyse / assembly_define failed: %s\n", mi355x_kkt_last_error(handle_));
            retval = SYMSOLVER_FATAL_ERROR;
            break;
         }
         for( int q = 0; q < NSEG; ++q )      // the sources filled so far go to the device
         {
            if( q == SEG_ID || seg_len_[q] == 0 )
            {
               continue;
            }
            std::memcpy(mi355x_kkt_assembly_buffer(handle_, q), &first_vals_[seg_off_[q]], sizeof(Number) * (size_t) seg_len_[q]);
            mi355x_kkt_assembly_upload(handle_, q);
            uploaded_bytes_ += 8ll * seg_len_[q];
         }
         std::vector<Number>().swap(first_vals_);
         // (irn_ / jcn_ stay: the primal-dual workspace may be asked for later)
         analysed_ = true;
         new_matrix = true;
      }
      else
      {
         const long long before = uploaded_bytes_;
         new_matrix = UpdateSources(W, W_factor, D_x, delta_x, D_s, delta_s, *J_c, D_c, delta_c, *J_d, D_d, delta_d, true);
         if( new_matrix && uploaded_bytes_ == before )
         {
            ++nfact_noupload_;
            Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X aug: refactoring with new perturbations only, no value upload\n");
         }
      }

      if( new_matrix || pivtol_changed_ )
      {
         if( HaveIpData() )
         {
            IpData().TimingStats().LinearSystemFactorization().Start();
         }
         int nneg = 0, nzero = 0;
         // (a pivot-tolerance change alone goes the same way: the sources are on the device)
         int st = mi355x_kkt_factor_assembled(handle_, scale_, shift_, &nneg, &nzero);
         if( HaveIpData() )
         {
            IpData().TimingStats().LinearSystemFactorization().End();
         }
         pivtol_changed_ = false;
         if( st == MI355X_KKT_FATAL )
         {
            Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_factor_assembled failed: %s\n", mi355x_kkt_last_error(handle_));
            retval = SYMSOLVER_FATAL_ERROR;
            have_factor_ = false;
            break;
         }
         have_factor_ = true;
         singular_ = (st == MI355X_KKT_SINGULAR);
         negevals_ = nneg;
         Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X aug factor: %d negative eigenvalues, %d zero pivots (status %d)\n", nneg, nzero, st);
         if( st == MI355X_KKT_SINGULAR )
         {
            retval = SYMSOLVER_SINGULAR;
            break;
         }
         if( check_NegEVals && negevals_ != numberOfNegEVals )
         {
            retval = SYMSOLVER_WRONG_INERTIA;
            break;
         }
      }
      else if( singular_ )
      {
         retval = SYMSOLVER_SINGULAR;      // the matrix has not changed: neither has the answer
         break;
      }

   }
   while( false );
   return retval;
}

ESymSolverStatus Mi355xAugSystemSolver::MultiSolve(const SymMatrix* W, Number W_factor, const Vector* D_x, Number delta_x,
      const Vector* D_s, Number delta_s, const Matrix* J_c, const Vector* D_c, Number delta_c, const Matrix* J_d,
      const Vector* D_d, Number delta_d, std::vector<SmartPtr<const Vector> >& rhs_xV, std::vector<SmartPtr<const Vector> >& rhs_sV,
      std::vector<SmartPtr<const Vector> >& rhs_cV, std::vector<SmartPtr<const Vector> >& rhs_dV, std::vector<SmartPtr<Vector> >& sol_xV,
      std::vector<SmartPtr<Vector> >& sol_sV, std::vector<SmartPtr<Vector> >& sol_cV, std::vector<SmartPtr<Vector> >& sol_dV,
      bool check_NegEVals, Index numberOfNegEVals)
{
   if( !J_c || !J_d )
   {
      return SYMSOLVER_FATAL_ERROR;    // as the reference: J_c and J_d MUST be given (IpStdAugSystemSolver.cpp:108)
   }
   if( HaveIpData() )
   {
      IpData().TimingStats().StdAugSystemSolverMultiSolve().Start();
   }
   const Index nrhs = (Index) rhs_xV.size();
   ESymSolverStatus retval = EnsureFactorization(W, W_factor, D_x, delta_x, D_s, delta_s, J_c, D_c, delta_c, J_d, D_d, delta_d,
                                                 check_NegEVals, numberOfNegEVals);
   do
   {
      if( retval != SYMSOLVER_SUCCESS )
      {
         break;
      }
      // right-hand sides: the four blocks of each system packed into one contiguous column (what CompoundVector +
      // TripletHelper::FillValuesFromVector do in TSymLinearSolver::MultiSolve, IpTSymLinearSolver.cpp:201-230)
      std::vector<Number> rhs((size_t) dim_ * nrhs);
      for( Index i = 0; i < nrhs; ++i )
      {
         Number* col = &rhs[(size_t) i * dim_];
         TripletHelper::FillValuesFromVector(n_x_, *rhs_xV[i], col);
         TripletHelper::FillValuesFromVector(n_s_, *rhs_sV[i], col + n_x_);
         TripletHelper::FillValuesFromVector(n_c_, *rhs_cV[i], col + n_x_ + n_s_);
         TripletHelper::FillValuesFromVector(n_d_, *rhs_dV[i], col + n_x_ + n_s_ + n_c_);
      }
      if( HaveIpData() )
      {
         IpData().TimingStats().LinearSystemBackSolve().Start();
      }
      int st = mi355x_kkt_solve(handle_, nrhs, &rhs[0], dim_);
      if( HaveIpData() )
      {
         IpData().TimingStats().LinearSystemBackSolve().End();
      }
      if( st != MI355X_KKT_SUCCESS )
      {
         Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_solve failed: %s\n", mi355x_kkt_last_error(handle_));
         retval = SYMSOLVER_FATAL_ERROR;
         break;
      }
      for( Index i = 0; i < nrhs; ++i )
      {
         const Number* col = &rhs[(size_t) i * dim_];
         TripletHelper::PutValuesInVector(n_x_, col, *sol_xV[i]);
         TripletHelper::PutValuesInVector(n_s_, col + n_x_, *sol_sV[i]);
         TripletHelper::PutValuesInVector(n_c_, col + n_x_ + n_s_, *sol_cV[i]);
         TripletHelper::PutValuesInVector(n_d_, col + n_x_ + n_s_ + n_c_, *sol_dV[i]);
      }
   }
   while( false );
   if( HaveIpData() )
   {
      IpData().TimingStats().StdAugSystemSolverMultiSolve().End();
   }
   return retval;
}

ESymSolverStatus Mi355xAugSystemSolver::Factorize(const SymMatrix* W, Number W_factor, const Vector* D_x, Number delta_x, const Vector* D_s,
      Number delta_s, const Matrix* J_c, const Vector* D_c, Number delta_c, const Matrix* J_d, const Vector* D_d, Number delta_d,
      bool check_NegEVals, Index numberOfNegEVals)
{
   if( !J_c || !J_d )
   {
      return SYMSOLVER_FATAL_ERROR;
   }
   if( HaveIpData() )
   {
      IpData().TimingStats().StdAugSystemSolverMultiSolve().Start();
   }
   const ESymSolverStatus retval = EnsureFactorization(W, W_factor, D_x, delta_x, D_s, delta_s, J_c, D_c, delta_c, J_d, D_d, delta_d,
                                   check_NegEVals, numberOfNegEVals);
   if( HaveIpData() )
   {
      IpData().TimingStats().StdAugSystemSolverMultiSolve().End();
   }
   return retval;
}

void Mi355xAugSystemSolver::WantPrimalDualWorkspace(const Index dims[8], const Index* idx_xl, const Index* idx_xu, const Index* idx_sl,
      const Index* idx_su)
{
   pd_wanted_ = true;
   for( int q = 0; q < 8; ++q )
   {
      pd_dims_[q] = dims[q];
   }
   const Index* src[4] = {idx_xl, idx_xu, idx_sl, idx_su};
   for( int q = 0; q < 4; ++q )
   {
      pd_idx_[q].assign(src[q], src[q] + dims[4 + q]);
   }
   if( analysed_ )     // the analysis has already happened (least-square multipliers come first): (re)define now -- pd_define frees an old workspace
   {
      DefinePrimalDualWorkspace();
   }
}

bool Mi355xAugSystemSolver::DefinePrimalDualWorkspace()
{
   // the primal-dual workspace reads W, J_c, J_d from the assembly sources (Mi355xPDSystemSolver)
   const int32_t segs[3] = {SEG_W, SEG_JC, SEG_JD};
   const int st = mi355x_kkt_pd_define(handle_, pd_dims_, pd_idx_[0].empty() ? NULL : &pd_idx_[0][0], pd_idx_[1].empty() ? NULL : &pd_idx_[1][0],
                                       pd_idx_[2].empty() ? NULL : &pd_idx_[2][0], pd_idx_[3].empty() ? NULL : &pd_idx_[3][0], &irn_[0], &jcn_[0], segs, 3);
   pd_defined_ = (st == MI355X_KKT_SUCCESS);
   if( !pd_defined_ )
   {
      Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_pd_define failed: %s\n", mi355x_kkt_last_error(handle_));
   }
   return pd_defined_;
}

bool Mi355xAugSystemSolver::IncreaseQuality()
{
   double unew = pivtol_;
   if( !handle_ || mi355x_kkt_increase_quality(handle_, &unew) == 0 )
   {
      return false;
   }
   Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "Increasing pivot tolerance for MI355X solver from %7.2e to %7.2e.\n", pivtol_, unew);
   pivtol_ = unew;
   pivtol_changed_ = true;
   return true;
}

SmartPtr<AlgorithmBuilder> MakeMi355xAugSystemAlgorithmBuilder()
{
   return new AlgorithmBuilder(new Mi355xAugSystemSolver(), "mi355x-ldlt (device-side KKT assembly)");
}

} // namespace Ipopt
