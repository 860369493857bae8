// ref_driver.cpp -- TEST / ORACLE INFRASTRUCTURE (never shipped, never on the product path).
//
// One executable that hosts the UNMODIFIED reference (oracle/_ref/libipopt_ref.so, built from
// /root/reference by oracle/ref_build.mk) and runs one of its own example NLPs
//   hs071                       reference examples/hs071_cpp/hs071_nlp.cpp
//   LukVlE1 / MBndryCntrl1 ...  reference examples/ScalableProblems/*.cpp (archive scalable.a)
// with either the reference's CPU linear solver (MKL PARDISO through the reference's own
// PardisoMKLSolverInterface -- the only CPU backend available offline) or the MI355X
// backend injected through the reference's virtual SymLinearSolverFactory (IpAlgBuilder.hpp:88).
//
// --record FILE wraps the chosen SparseSymLinearSolverInterface in a decorator that stores every
// call crossing the boundary (structure, values, rhs, returned status / inertia / solution).  That
// is how tests/golden/*.kktrec are produced from the reference itself (tests/golden/make_golden.sh).
//
// --solver stock leaves the choice to the reference's own AlgorithmBuilder, i.e. to the `linear_solver` option
// (--set linear_solver ma97 --set hsllib .../libmi355x_kkt.so = route B2; --set linear_solver mi355x with the patched
// library oracle/_ref/libipopt_ref_mi355x.so = route B1').
//
// usage: ref_driver <problem> <N> [--solver pardisomkl|mi355x|mi355x-aug|mi355x-pd|stock] [--record file] [--max-records K] [--skip-records K]
//                   [--set name value]... [--optfile ipopt.opt] [--reoptimize] [--quiet]
//   --reoptimize: after the first solve, set warm_start_same_structure=yes and call ReOptimizeNLP (second DRIVER_SUMMARY line)
//   --then-bounds lo hi: (LukVl problems) after the first solve, optimise a SECOND instance of the problem with constraint bounds [lo, hi]
//                   through the SAME AlgorithmBuilder, without warm_start_same_structure: other bound sets, other workspace dimensions
#include "IpIpoptApplication.hpp"
#include "IpTNLPAdapter.hpp"
#include "IpAlgBuilder.hpp"
#include "IpTSymLinearSolver.hpp"
#include "IpSparseSymLinearSolverInterface.hpp"
#include "IpPardisoMKLSolverInterface.hpp"
#include "IpIpoptData.hpp"
#include "IpTimingStatistics.hpp"
#include "IpSolveStatistics.hpp"
#include "hs071_nlp.hpp"
#include "LuksanVlcek1.hpp"
#include "LuksanVlcek2.hpp"
#include "LuksanVlcek3.hpp"
#include "LuksanVlcek4.hpp"
#include "LuksanVlcek5.hpp"
#include "LuksanVlcek6.hpp"
#include "LuksanVlcek7.hpp"
#include "MittelmannBndryCntrlDiri.hpp"
#include "MittelmannBndryCntrlDiri3D.hpp"
#include "MittelmannBndryCntrlDiri3D_27.hpp"
#include "MittelmannBndryCntrlDiri3Dsin.hpp"
#include "MittelmannBndryCntrlNeum.hpp"
#include "MittelmannDistCntrlDiri.hpp"
#include "MittelmannDistCntrlNeumA.hpp"
#include "MittelmannDistCntrlNeumB.hpp"
#include "MittelmannParaCntrl.hpp"
#ifdef WITH_MI355X
#include "IpMi355xSolverInterface.hpp"
#include "IpMi355xAugSystemSolver.hpp"
#include "IpMi355xPDSystemSolver.hpp"
#endif
#include "IpPDSystemSolver.hpp"
#include "IpTripletHelper.hpp"
#include "IpExpansionMatrix.hpp"
#include "IpIpoptCalculatedQuantities.hpp"
#include "IpIpoptData.hpp"
#include "IpIpoptNLP.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>

using namespace Ipopt;

// A 6-variable test NLP of this harness (not from the reference): min sum (x_i - i)^2  s.t. three equality constraints of which
// the third is the sum of the first two -- exercises dependency_detector (TNLPAdapter::DetermineDependentConstraints,
// IpTNLPAdapter.cpp:636-700 -> TSymDependencyDetector -> SparseSymLinearSolverInterface::DetermineDependentRows).
class DepTestNLP: public TNLP
{
public:
   bool get_nlp_info(Index& n, Index& m, Index& nnz_jac_g, Index& nnz_h_lag, IndexStyleEnum& index_style)
   { n = 6; m = 3; nnz_jac_g = 8; nnz_h_lag = 6; index_style = C_STYLE; return true; }
   bool get_bounds_info(Index n, Number* x_l, Number* x_u, Index m, Number* g_l, Number* g_u)
   { for( Index i = 0; i < n; ++i ) { x_l[i] = -1e19; x_u[i] = 1e19; } g_l[0] = g_u[0] = 1.; g_l[1] = g_u[1] = 2.; g_l[2] = g_u[2] = 3.; return true; }
   bool get_starting_point(Index n, bool, Number* x, bool, Number*, Number*, Index, bool, Number*)
   { for( Index i = 0; i < n; ++i ) x[i] = 0.; return true; }
   bool eval_f(Index n, const Number* x, bool, Number& f)
   { f = 0.; for( Index i = 0; i < n; ++i ) f += (x[i] - (i + 1)) * (x[i] - (i + 1)); return true; }
   bool eval_grad_f(Index n, const Number* x, bool, Number* g)
   { for( Index i = 0; i < n; ++i ) g[i] = 2. * (x[i] - (i + 1)); return true; }
   bool eval_g(Index, const Number* x, bool, Index, Number* g)
   { g[0] = x[0] + x[1]; g[1] = x[2] + x[3]; g[2] = x[0] + x[1] + x[2] + x[3]; return true; }
   bool eval_jac_g(Index, const Number*, bool, Index, Index, Index* iRow, Index* jCol, Number* v)
   {
      static const Index r[8] = {0, 0, 1, 1, 2, 2, 2, 2}, c[8] = {0, 1, 2, 3, 0, 1, 2, 3};
      if( !v ) for( int k = 0; k < 8; ++k ) { iRow[k] = r[k]; jCol[k] = c[k]; }
      else for( int k = 0; k < 8; ++k ) v[k] = 1.;
      return true;
   }
   bool eval_h(Index n, const Number*, bool, Number obj_factor, Index, const Number*, bool, Index, Index* iRow, Index* jCol, Number* v)
   {
      if( !v ) for( Index i = 0; i < n; ++i ) { iRow[i] = jCol[i] = i; }
      else for( Index i = 0; i < n; ++i ) v[i] = 2. * obj_factor;
      return true;
   }
   void finalize_solution(SolverReturn, Index, const Number*, const Number*, const Number*, Index, const Number*, const Number*, Number,
                          const IpoptData*, IpoptCalculatedQuantities*)
   { }
};

// ------------------------------------------------------------------------------------------
// recording decorator
// ------------------------------------------------------------------------------------------
class RecordingSolverInterface: public SparseSymLinearSolverInterface
{
public:
   RecordingSolverInterface(SmartPtr<SparseSymLinearSolverInterface> inner, const std::string& file, int max_records, int skip_records = 0)
      : inner_(inner), f_(NULL), dim_(0), nnz_(0), nrec_(0), max_records_(max_records), ncalls_(0), skip_records_(skip_records), vals_(NULL)
   {
      f_ = fopen(file.c_str(), "wb");
      if( f_ ) fwrite("KKTREC1\n", 1, 8, f_);
   }
   ~RecordingSolverInterface() { if( f_ ) fclose(f_); printf("RECORD_CALLS %d\n", ncalls_); }
   bool InitializeImpl(const OptionsList& options, const std::string& prefix)
   {
      return inner_->Initialize(Jnlst(), IpNLP(), IpData(), IpCq(), options, prefix);
   }
   ESymSolverStatus InitializeStructure(Index dim, Index nonzeros, const Index* ia, const Index* ja)
   {
      dim_ = dim; nnz_ = nonzeros;
      int fmt = (int) inner_->MatrixFormat();
      Index nia = (fmt == (int) Triplet_Format) ? nonzeros : dim + 1;
      if( f_ )
      {
         int hdr[8] = {0, dim, nonzeros, fmt, (int) nia, 0, 0, 0};
         fwrite(hdr, sizeof(int), 8, f_);
         fwrite(ia, sizeof(Index), nia, f_);
         fwrite(ja, sizeof(Index), nonzeros, f_);
      }
      return inner_->InitializeStructure(dim, nonzeros, ia, ja);
   }
   Number* GetValuesArrayPtr() { vals_ = inner_->GetValuesArrayPtr(); return vals_; }
   ESymSolverStatus MultiSolve(bool new_matrix, const Index* ia, const Index* ja, Index nrhs, Number* rhs_vals,
                               bool check_NegEVals, Index numberOfNegEVals)
   {
      std::vector<Number> a, rhs(rhs_vals, rhs_vals + (size_t) dim_ * nrhs);
      // NB: some backends (MKL adapter) may overwrite their value array; copy before the call
      if( new_matrix && vals_ ) a.assign(vals_, vals_ + nnz_);
      ESymSolverStatus st = inner_->MultiSolve(new_matrix, ia, ja, nrhs, rhs_vals, check_NegEVals, numberOfNegEVals);
      // --skip-records K: the first K calls are not stored (the LATE calls of a run -- Sigma spanning many decades, delta_c active -- are what
      // is wanted); the first call that IS stored carries the matrix it was answered with even if that matrix is not new to the backend
      if( skip_records_ > 0 && new_matrix ) last_a_ = a;
      const bool store = f_ && ncalls_ >= skip_records_ && (max_records_ < 0 || nrec_ < max_records_);
      ++ncalls_;
      if( store && !new_matrix && nrec_ == 0 && skip_records_ > 0 ) { a = last_a_; new_matrix = true; }
      if( store )
      {
         int hdr[8] = {1, dim_, nnz_, nrhs, new_matrix ? 1 : 0, check_NegEVals ? 1 : 0, numberOfNegEVals, (int) st};
         fwrite(hdr, sizeof(int), 8, f_);
         int neg = inner_->ProvidesInertia() ? inner_->NumberOfNegEVals() : -1;
         fwrite(&neg, sizeof(int), 1, f_);
         if( new_matrix ) fwrite(a.data(), sizeof(Number), nnz_, f_);
         fwrite(rhs.data(), sizeof(Number), (size_t) dim_ * nrhs, f_);
         fwrite(rhs_vals, sizeof(Number), (size_t) dim_ * nrhs, f_);
         ++nrec_;
      }
      return st;
   }
   Index NumberOfNegEVals() const { return inner_->NumberOfNegEVals(); }
   bool IncreaseQuality() { return inner_->IncreaseQuality(); }
   bool ProvidesInertia() const { return inner_->ProvidesInertia(); }
   EMatrixFormat MatrixFormat() const { return inner_->MatrixFormat(); }
private:
   SmartPtr<SparseSymLinearSolverInterface> inner_;
   FILE* f_;
   Index dim_, nnz_;
   int nrec_, max_records_, ncalls_, skip_records_;
   std::vector<Number> last_a_;
   Number* vals_;
};


// ------------------------------------------------------------------------------------------
// recording decorator of the PDSystemSolver boundary: what PDFullSpaceSolver::Solve was given -- the pieces of the
// 8-block system of the current iterate and the right-hand side -- and what it returned.  Golden fixtures for oracle/pd_oracle.py.
//   file: "PDREC1\n", then per call: int hdr[16] = {nx, ns, nc, nd, nxl, nxu, nsl, nsu, nnzW, nnzJc, nnzJd, allow_inexact, improve_solution, ok, 0, 0},
//   double {alpha, beta, delta_x, delta_s, delta_c, delta_d}; int index lists of the 4 expansion matrices (0-based); int (irow, jcol) 1-based +
//   double values of W, J_c, J_d; double z_L, z_U, v_L, v_U, slack_x_L, slack_x_U, slack_s_L, slack_s_U, sigma_x, sigma_s;
//   double rhs (8 blocks), res on entry (8 blocks), res on return (8 blocks)
// ------------------------------------------------------------------------------------------
class RecordingPDSolver: public PDSystemSolver
{
public:
   RecordingPDSolver(SmartPtr<PDSystemSolver> inner, const std::string& file, int max_records)
      : inner_(inner), f_(NULL), nrec_(0), max_records_(max_records)
   {
      f_ = fopen(file.c_str(), "wb");
      if( f_ ) fwrite("PDREC1\n", 1, 7, f_);
   }
   ~RecordingPDSolver() { if( f_ ) fclose(f_); }
   bool InitializeImpl(const OptionsList& options, const std::string& prefix)
   {
      return inner_->Initialize(Jnlst(), IpNLP(), IpData(), IpCq(), options, prefix);
   }
   bool Solve(Number alpha, Number beta, const IteratesVector& rhs, IteratesVector& res, bool allow_inexact, bool improve_solution)
   {
      const bool rec = f_ && (max_records_ < 0 || nrec_ < max_records_);
      std::vector<Number> res_in;
      if( rec && (beta != 0. || improve_solution) ) Blocks(res, res_in);      // (otherwise `res` may be uninitialised memory)
      const bool ok = inner_->Solve(alpha, beta, rhs, res, allow_inexact, improve_solution);
      if( rec )
      {
         SmartPtr<const SymMatrix> W = IpData().W();
         SmartPtr<const Matrix> Jc = IpCq().curr_jac_c(), Jd = IpCq().curr_jac_d();
         const ExpansionMatrix* P[4] = {dynamic_cast<const ExpansionMatrix*>(GetRawPtr(IpNLP().Px_L())), dynamic_cast<const ExpansionMatrix*>(GetRawPtr(IpNLP().Px_U())),
                                        dynamic_cast<const ExpansionMatrix*>(GetRawPtr(IpNLP().Pd_L())), dynamic_cast<const ExpansionMatrix*>(GetRawPtr(IpNLP().Pd_U()))};
         if( P[0] && P[1] && P[2] && P[3] )
         {
            const Index nW = TripletHelper::GetNumberEntries(*W), nJc = TripletHelper::GetNumberEntries(*Jc), nJd = TripletHelper::GetNumberEntries(*Jd);
            int hdr[16] = {rhs.x()->Dim(), rhs.s()->Dim(), rhs.y_c()->Dim(), rhs.y_d()->Dim(), P[0]->NCols(), P[1]->NCols(), P[2]->NCols(), P[3]->NCols(),
                           nW, nJc, nJd, allow_inexact ? 1 : 0, improve_solution ? 1 : 0, ok ? 1 : 0, 0, 0};
            fwrite(hdr, sizeof(int), 16, f_);
            Number dx, ds, dc, dd;
            IpData().getPDPert(dx, ds, dc, dd);
            double sc[6] = {alpha, beta, dx, ds, dc, dd};
            fwrite(sc, sizeof(double), 6, f_);
            for( int q = 0; q < 4; ++q ) if( P[q]->NCols() > 0 ) fwrite(P[q]->ExpandedPosIndices(), sizeof(Index), P[q]->NCols(), f_);
            Trip(nW, *W); Trip(nJc, *Jc); Trip(nJd, *Jd);
            const Vector* dat[10] = {GetRawPtr(IpData().curr()->z_L()), GetRawPtr(IpData().curr()->z_U()), GetRawPtr(IpData().curr()->v_L()), GetRawPtr(IpData().curr()->v_U()),
                                     GetRawPtr(IpCq().curr_slack_x_L()), GetRawPtr(IpCq().curr_slack_x_U()), GetRawPtr(IpCq().curr_slack_s_L()), GetRawPtr(IpCq().curr_slack_s_U()),
                                     GetRawPtr(IpCq().curr_sigma_x()), GetRawPtr(IpCq().curr_sigma_s())};
            for( int q = 0; q < 10; ++q ) Vec(*dat[q]);
            std::vector<Number> b;
            Blocks(rhs, b); fwrite(b.data(), sizeof(Number), b.size(), f_);
            if( res_in.empty() ) res_in.assign(b.size(), 0.);
            fwrite(res_in.data(), sizeof(Number), res_in.size(), f_);
            Blocks(res, b); fwrite(b.data(), sizeof(Number), b.size(), f_);
            ++nrec_;
         }
      }
      return ok;
   }
private:
   void Vec(const Vector& v)
   {
      std::vector<Number> a(v.Dim() > 0 ? v.Dim() : 1);
      if( v.Dim() > 0 ) { TripletHelper::FillValuesFromVector(v.Dim(), v, a.data()); fwrite(a.data(), sizeof(Number), v.Dim(), f_); }
   }
   void Trip(Index n, const Matrix& M)
   {
      if( n <= 0 ) return;
      std::vector<Index> r(n), c(n); std::vector<Number> v(n);
      TripletHelper::FillRowCol(n, M, r.data(), c.data());
      TripletHelper::FillValues(n, M, v.data());
      fwrite(r.data(), sizeof(Index), n, f_); fwrite(c.data(), sizeof(Index), n, f_); fwrite(v.data(), sizeof(Number), n, f_);
   }
   static void Blocks(const IteratesVector& it, std::vector<Number>& out)
   {
      const Vector* b[8] = {GetRawPtr(it.x()), GetRawPtr(it.s()), GetRawPtr(it.y_c()), GetRawPtr(it.y_d()), GetRawPtr(it.z_L()), GetRawPtr(it.z_U()), GetRawPtr(it.v_L()), GetRawPtr(it.v_U())};
      out.clear();
      for( int q = 0; q < 8; ++q )
      {
         const size_t o = out.size();
         out.resize(o + b[q]->Dim());
         if( b[q]->Dim() > 0 ) TripletHelper::FillValuesFromVector(b[q]->Dim(), *b[q], &out[o]);
      }
   }
   SmartPtr<PDSystemSolver> inner_;
   FILE* f_;
   int nrec_, max_records_;
};

class DriverAlgBuilder: public AlgorithmBuilder
{
public:
   DriverAlgBuilder(const std::string& solver, const std::string& record, int max_records, const std::string& record_pd = "", int skip_records = 0)
      : solver_(solver), record_(record), record_pd_(record_pd), max_records_(max_records), skip_records_(skip_records) { }
   virtual SmartPtr<PDSystemSolver> PDSystemSolverFactory(const Journalist& jnlst, const OptionsList& options, const std::string& prefix)
   {
      SmartPtr<PDSystemSolver> pd = AlgorithmBuilder::PDSystemSolverFactory(jnlst, options, prefix);      // the reference's PDFullSpaceSolver
      if( !record_pd_.empty() ) pd = new RecordingPDSolver(pd, record_pd_, max_records_);
      return pd;
   }
   virtual SmartPtr<SymLinearSolver> SymLinearSolverFactory(const Journalist& jnlst, const OptionsList& options, const std::string& prefix)
   {
#ifdef WITH_MI355X
      // the product's own builder (honours mi355x_outer_scaling) unless the boundary is being recorded
      if( solver_ == "mi355x" && record_.empty() ) return Mi355xAlgorithmBuilder().SymLinearSolverFactory(jnlst, options, prefix);
#endif
      SmartPtr<SparseSymLinearSolverInterface> iface;
      if( solver_ == "pardisomkl" ) iface = new PardisoMKLSolverInterface();
#ifdef WITH_MI355X
      else if( solver_ == "mi355x" ) iface = new Mi355xSolverInterface();
#endif
      else { fprintf(stderr, "unknown --solver %s\n", solver_.c_str()); exit(2); }
      if( !record_.empty() ) iface = new RecordingSolverInterface(iface, record_, max_records_, skip_records_);
      SmartPtr<TSymScalingMethod> none;
      return new TSymLinearSolver(iface, none);
   }
private:
   std::string solver_, record_, record_pd_;
   int max_records_, skip_records_;
};

static double wall(const TimedTask& t) { return t.TotalWallclockTime(); }

int main(int argc, char** argv)
{
   if( argc < 3 ) { fprintf(stderr, "usage: %s <problem> <N> [--solver s] [--record f] [--max-records k] [--set k v]... [--quiet]\n", argv[0]); return 2; }
   std::string problem = argv[1];
   int N = atoi(argv[2]);
   std::string solver = "pardisomkl", record, record_pd;
   int max_records = -1, skip_records = 0;
   bool quiet = false, reopt = false, then_bounds = false;
   double tb_lo = 0., tb_hi = 0.;
   std::string optfile;
   std::vector<std::pair<std::string, std::string> > sets;
   for( int i = 3; i < argc; ++i )
   {
      std::string a = argv[i];
      if( a == "--solver" && i + 1 < argc ) solver = argv[++i];
      else if( a == "--record" && i + 1 < argc ) record = argv[++i];
      else if( a == "--record-pd" && i + 1 < argc ) record_pd = argv[++i];
      else if( a == "--max-records" && i + 1 < argc ) max_records = atoi(argv[++i]);
      else if( a == "--skip-records" && i + 1 < argc ) skip_records = atoi(argv[++i]);
      else if( a == "--set" && i + 2 < argc ) { sets.push_back(std::make_pair(std::string(argv[i + 1]), std::string(argv[i + 2]))); i += 2; }
      else if( a == "--quiet" ) quiet = true;
      else if( a == "--reoptimize" ) reopt = true;
      else if( a == "--then-bounds" && i + 2 < argc ) { then_bounds = true; tb_lo = atof(argv[i + 1]); tb_hi = atof(argv[i + 2]); i += 2; }
      else if( a == "--optfile" && i + 1 < argc ) optfile = argv[++i];
      else { fprintf(stderr, "bad argument %s\n", a.c_str()); return 2; }
   }

   SmartPtr<TNLP> tnlp;
   if( problem == "hs071" ) tnlp = new HS071_NLP();
   else if( problem == "deptest" ) tnlp = new DepTestNLP();
   else
   {
      SmartPtr<RegisteredTNLP> r;
      // every problem class the reference's own driver registers (examples/ScalableProblems/solve_problem.cpp:28-91), under the same names
#define PROB(name, ctor) else if( problem == #name ) r = new ctor
      if( problem == "LukVlI1u" ) r = new LuksanVlcek1(-1., 1e20);      // (ours: LukVlI1 with the upper constraint bound removed)
      PROB(LukVlE1, LuksanVlcek1(0, 0)); PROB(LukVlI1, LuksanVlcek1(-1., 0.));
      PROB(LukVlE2, LuksanVlcek2(0, 0)); PROB(LukVlI2, LuksanVlcek2(-1., 0.));
      PROB(LukVlE3, LuksanVlcek3(0, 0)); PROB(LukVlI3, LuksanVlcek3(-1., 0.));
      PROB(LukVlE4, LuksanVlcek4(0, 0)); PROB(LukVlI4, LuksanVlcek4(-1., 0.));
      PROB(LukVlE5, LuksanVlcek5(0, 0)); PROB(LukVlI5, LuksanVlcek5(-1., 0.));
      PROB(LukVlE6, LuksanVlcek6(0, 0)); PROB(LukVlI6, LuksanVlcek6(-1., 0.));
      PROB(LukVlE7, LuksanVlcek7(0, 0)); PROB(LukVlI7, LuksanVlcek7(-1., 0.));
      PROB(MBndryCntrl1, MittelmannBndryCntrlDiri1()); PROB(MBndryCntrl2, MittelmannBndryCntrlDiri2());
      PROB(MBndryCntrl3, MittelmannBndryCntrlDiri3()); PROB(MBndryCntrl4, MittelmannBndryCntrlDiri4());
      PROB(MBndryCntrl_3D, MittelmannBndryCntrlDiri3D()); PROB(MBndryCntrl_3D_27, MittelmannBndryCntrlDiri3D_27());
      PROB(MBndryCntrl_3D_27BT, MittelmannBndryCntrlDiri3D_27BT()); PROB(MBndryCntrl_3Dsin, MittelmannBndryCntrlDiri3Dsin());
      PROB(MBndryCntrl5, MittelmannBndryCntrlNeum1()); PROB(MBndryCntrl6, MittelmannBndryCntrlNeum2());
      PROB(MBndryCntrl7, MittelmannBndryCntrlNeum3()); PROB(MBndryCntrl8, MittelmannBndryCntrlNeum4());
      PROB(MDistCntrl1, MittelmannDistCntrlDiri1()); PROB(MDistCntrl2, MittelmannDistCntrlDiri2());
      PROB(MDistCntrl3, MittelmannDistCntrlDiri3()); PROB(MDistCntrl3a, MittelmannDistCntrlDiri3a());
      PROB(MDistCntrl4, MittelmannDistCntrlNeumA1()); PROB(MDistCntrl5, MittelmannDistCntrlNeumA2()); PROB(MDistCntrl6a, MittelmannDistCntrlNeumA3());
      PROB(MDistCntrl4a, MittelmannDistCntrlNeumB1()); PROB(MDistCntrl5a, MittelmannDistCntrlNeumB2()); PROB(MDistCntrl6, MittelmannDistCntrlNeumB3());
      PROB(MPara5_1, MittelmannParaCntrlBase<MittelmannParaCntrl5_1>()); PROB(MPara5_2_1, MittelmannParaCntrlBase<MittelmannParaCntrl5_2_1>());
      PROB(MPara5_2_2, MittelmannParaCntrlBase<MittelmannParaCntrl5_2_2>()); PROB(MPara5_2_3, MittelmannParaCntrlBase<MittelmannParaCntrl5_2_3>());
#undef PROB
      else { fprintf(stderr, "unknown problem %s\n", problem.c_str()); return 2; }
      if( !r->InitializeProblem(N) ) { fprintf(stderr, "InitializeProblem(%d) failed\n", N); return 2; }
      tnlp = GetRawPtr(r);
   }

   SmartPtr<IpoptApplication> app = IpoptApplicationFactory();
#ifdef WITH_MI355X
   Mi355xSolverInterface::RegisterOptions(app->RegOptions());   // (the patched library registers them itself, IpLinearSolversRegOp.cpp)
#endif
   app->Options()->SetStringValue("print_timing_statistics", "yes");
   if( problem == "hs071" )
   {  // the settings of the reference's own test driver, examples/hs071_cpp/hs071_main.cpp:33-35
      app->Options()->SetNumericValue("tol", 3.82e-6);
      app->Options()->SetStringValue("mu_strategy", "adaptive");
   }
   if( quiet ) app->Options()->SetIntegerValue("print_level", 0);
   for( size_t i = 0; i < sets.size(); ++i )
   {
      const std::string& k = sets[i].first; const std::string& v = sets[i].second;
      char* end = NULL; double d = strtod(v.c_str(), &end);
      bool isnum = end && *end == 0 && !v.empty();
      bool ok = false;
      if( isnum && v.find_first_of(".eE") == std::string::npos ) ok = app->Options()->SetIntegerValue(k, atoi(v.c_str()), true, true);
      if( !ok && isnum ) ok = app->Options()->SetNumericValue(k, d, true, true);
      if( !ok ) ok = app->Options()->SetStringValue(k, v, true, true);
      if( !ok ) { fprintf(stderr, "could not set option %s=%s\n", k.c_str(), v.c_str()); return 2; }
   }
   if( app->Initialize(optfile) != Solve_Succeeded ) { fprintf(stderr, "Initialize failed\n"); return 3; }

   SmartPtr<NLP> nlp = new TNLPAdapter(tnlp, app->Jnlst());
   SmartPtr<AlgorithmBuilder> builder;
#ifdef WITH_MI355X
   SmartPtr<Mi355xAugSystemSolver> aug;
#endif
   if( solver == "stock" ) builder = new AlgorithmBuilder();     // the reference's own factory chain (IpAlgBuilder.cpp:427-526)
#ifdef WITH_MI355X
   else if( solver == "mi355x-aug" )
   {  // route (ii): custom AugSystemSolver with device-side KKT assembly; needs linear_solver=custom (IpAlgBuilder.cpp:576-584)
      app->Options()->SetStringValue("linear_solver", "custom");
      aug = new Mi355xAugSystemSolver();
      builder = new AlgorithmBuilder(GetRawPtr(aug), "mi355x-ldlt (device-side KKT assembly)");
   }
   else if( solver == "mi355x-pd" )
   {  // the full device route: custom AugSystemSolver + PDSystemSolver with the primal-dual vectors resident on the GPU
      app->Options()->SetStringValue("linear_solver", "custom");
      builder = MakeMi355xPDSystemAlgorithmBuilder();
   }
#endif
   else builder = new DriverAlgBuilder(solver, record, max_records, record_pd, skip_records);
   auto t0 = std::chrono::steady_clock::now();
   ApplicationReturnStatus status = app->OptimizeNLP(nlp, builder);
   double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

   for( int pass = 0; pass < ((reopt || then_bounds) ? 2 : 1); ++pass )
   {
   if( pass == 1 && then_bounds )
   {  // another instance (other bounds => other dimensions of the bound blocks) through the SAME builder, no warm start: everything the
      // augmented-system solver remembers of the first problem must go
      SmartPtr<RegisteredTNLP> r2 = new LuksanVlcek1(tb_lo, tb_hi);
      if( !r2->InitializeProblem(N) ) return 2;
      SmartPtr<NLP> nlp2 = new TNLPAdapter(GetRawPtr(r2), app->Jnlst());
      t0 = std::chrono::steady_clock::now();
      status = app->OptimizeNLP(nlp2, builder);
      total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   }
   else if( pass == 1 )
   {  // same structure, same start: the backend must keep its symbolic analysis (IpMumpsSolverInterface.cpp:227-236 pattern)
      app->Options()->SetStringValue("warm_start_same_structure", "yes");
      t0 = std::chrono::steady_clock::now();
      status = app->ReOptimizeNLP(nlp);
      total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   }
   SmartPtr<SolveStatistics> stats = app->Statistics();
   int iters = IsValid(stats) ? stats->IterationCount() : -1;
   double obj = IsValid(stats) ? stats->FinalObjective() : 0.;
   TimingStatistics& ts = app->IpoptDataObject()->TimingStats();
   printf("DRIVER_SUMMARY {\"problem\": \"%s\", \"N\": %d, \"solver\": \"%s\", \"status\": %d, \"iterations\": %d, \"objective\": %.16e, "
          "\"wall_total\": %.6f, \"PDSystemSolverTotal\": %.6f, \"PDSystemSolverSolveOnce\": %.6f, \"LinearSystemFactorization\": %.6f, "
          "\"LinearSystemBackSolve\": %.6f, \"LinearSystemSymbolicFactorization\": %.6f, \"LinearSystemStructureConverter\": %.6f, "
          "\"StdAugSystemSolverMultiSolve\": %.6f, \"OverallAlgorithm\": %.6f, \"TotalFunctionEvaluations\": %.6f}\n",
          problem.c_str(), N, solver.c_str(), (int) status, iters, obj, total,
          wall(ts.PDSystemSolverTotal()), wall(ts.PDSystemSolverSolveOnce()), wall(ts.LinearSystemFactorization()),
          wall(ts.LinearSystemBackSolve()), wall(ts.LinearSystemSymbolicFactorization()), wall(ts.LinearSystemStructureConverter()),
          wall(ts.StdAugSystemSolverMultiSolve()), wall(ts.OverallAlgorithm()), (double) ts.TotalFunctionEvaluationWallclockTime());
   }
#ifdef WITH_MI355X
   if( IsValid(aug) )
      printf("AUG_STATS {\"uploaded_value_bytes\": %lld, \"factorizations_without_upload\": %d}\n", aug->UploadedBytes(), (int) aug->FactorizationsWithoutUpload());
   {
      Index ndev = 0, nhost = 0, nref = 0;
      if( GetMi355xPDSystemStatistics(builder, ndev, nhost, nref) )
         printf("PD_STATS {\"device_solves\": %d, \"host_solves\": %d, \"refinement_steps\": %d}\n", (int) ndev, (int) nhost, (int) nref);
   }
#endif
   return (status == Solve_Succeeded || status == Solved_To_Acceptable_Level) ? 0 : 1;
}
