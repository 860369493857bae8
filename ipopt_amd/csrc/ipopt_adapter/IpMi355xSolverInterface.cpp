// IpMi355xSolverInterface.cpp -- see the header.  Status protocol as in IpSymLinearSolver.hpp:19-33.
#include "IpMi355xSolverInterface.hpp"
#include "IpMi355xTSymScalingMethod.hpp"
#include "IpMi355xCommBootstrap.hpp"
#include "IpTSymLinearSolver.hpp"
#include "IpIpoptData.hpp"
#include "IpTimingStatistics.hpp"
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <string>
#include <unistd.h>
#include <fcntl.h>
#include <ctime>
#include <sys/stat.h>
#include <sys/types.h>

namespace Ipopt
{

Mi355xSolverInterface::Mi355xSolverInterface()
   : handle_(NULL), dim_(0), nonzeros_(0), ia_(NULL), ja_(NULL), analysed_(false), pivtol_changed_(false),
     warm_start_same_structure_(false), pivtol_(1e-8), pivtolmax_(1e-4), negevals_(-1)
{
   mi355x_kkt_default_options(&kopts_);
}

Mi355xSolverInterface::~Mi355xSolverInterface()
{
   if( handle_ )
   {
      mi355x_kkt_destroy(handle_);
   }
}

void Mi355xSolverInterface::RegisterOptions(SmartPtr<RegisteredOptions> roptions)
{
   roptions->SetRegisteringCategory("MI355X Linear Solver");
   roptions->AddBoundedNumberOption("mi355x_pivtol", "Pivot tolerance for the MI355X LDL^T solver.", 0., true, 0.5,
                                    false, 1e-8, "Relative threshold u of the 1x1 / 2x2 pivot tests (MA27/MA57 tests against the whole front column).");
   roptions->AddBoundedNumberOption("mi355x_pivtolmax", "Maximum pivot tolerance for the MI355X LDL^T solver.", 0.,
                                    true, 0.5, false, 1e-4, "IncreaseQuality raises the tolerance u <- u^0.75 up to this value.");
   roptions->AddStringOption6("mi355x_scaling", "Symmetric scaling of the KKT matrix inside the backend.", "ruiz", "none",
                              "no scaling", "ruiz", "4 sweeps of inf-norm Ruiz equilibration on the device (the algorithm of MC77)", "matching",
                              "maximum-product matching scaling (the job of MC64), computed on the host at the first factorisation and reused until IncreaseQuality (ma97_switch ...-reuse)", "matching-always",
                              "the same, recomputed at every factorisation", "matching-device",
                              "maximum-product matching scaling computed ON THE DEVICE (Jacobi auction), at the first factorisation and reused until IncreaseQuality", "matching-device-always",
                              "the same, recomputed at every factorisation");
   roptions->AddStringOption3("mi355x_ordering", "Fill-reducing ordering.", "nd", "nd",
                              "nested dissection with minimum-degree leaves", "md", "minimum degree", "natural", "identity");
   roptions->AddStringOption2("mi355x_matching", "Pre-pair zero-diagonal rows into 2x2-capable supernodes.", "yes", "no", "",
                              "yes", "");
   roptions->AddStringOption3("mi355x_outer_scaling", "Scale through Ipopt's TSymScalingMethod hook instead of inside the backend.", "no",
                              "no", "the backend scales internally (mi355x_scaling)", "yes",
                              "Ruiz factors computed on the device are applied by TSymLinearSolver, subject to linear_scaling_on_demand", "matching",
                              "same with maximum-product matching (MC64-style) factors");
   roptions->AddLowerBoundedIntegerOption("mi355x_nemin", "Supernode amalgamation parameter.", 1, 8, "");
   roptions->AddLowerBoundedIntegerOption("mi355x_nd_leaf", "Nested dissection leaf size.", 8, 32, "");
   roptions->AddLowerBoundedIntegerOption("mi355x_max_sn_cols", "Maximum columns per supernode.", 2, 64, "");
   roptions->AddLowerBoundedIntegerOption("mi355x_device", "HIP device ordinal (-1: current).", -1, -1, "");
   roptions->AddLowerBoundedIntegerOption("mi355x_verbose", "Verbosity of the MI355X backend.", 0, 0, "");
   // multi-GPU: one Ipopt process per GPU, every process runs the same algorithm, the KKT factorisation is shared
   // (elimination-tree subtrees per rank, RCCL all-reduce at the subtree joins); cf. the SPRAL knobs IpSpralSolverInterface.cpp:55-67
   roptions->AddLowerBoundedIntegerOption("mi355x_delay_rounds", "Delayed-pivot rounds per factorisation.", 0, 8,
                                          "Columns that fail the pivot threshold in their front are moved to the parent front and the matrix is refactored, "
                                          "at most this many times per factorisation (what MA27 / MA57 / MA97 / MUMPS do inside one call); 0 = static pivoting.");
   roptions->AddStringOption2("mi355x_smart_quality", "IncreaseQuality answers 'no' when the last factorisation did not depend on the pivot tolerance.", "no",
                              "no", "raise the pivot tolerance whenever it is below mi355x_pivtolmax (what the MA27 / MA57 / MA97 adapters do)",
                              "yes", "skip the refactorisation when no pivot decision would change up to mi355x_pivtolmax");
   roptions->AddLowerBoundedIntegerOption("mi355x_nranks", "Number of processes (GPUs) sharing each KKT factorisation.", 0, 0,
                                          "0: take WORLD_SIZE / OMPI_COMM_WORLD_SIZE from the environment (1 if unset).");
   roptions->AddLowerBoundedIntegerOption("mi355x_rank", "Rank of this process among mi355x_nranks.", -1, -1,
                                          "-1: take RANK / OMPI_COMM_WORLD_RANK from the environment.");
   roptions->AddStringOption2("mi355x_subcube", "Fronts above the ranks' subtrees are replicated only on the ranks beneath them.", "no",
                              "no", "one top of the elimination tree replicated on every rank, one exchange step per factorisation",
                              "yes", "subtree-to-subcube mapping: one exchange step per bisection of the machine (more than two ranks)",
                              "Multi-GPU partition of the KKT factorisation (load-balance knob; cf. IpSpralSolverInterface.cpp:55-67).");
   roptions->AddStringOption2("mi355x_comm", "Communicator among the mi355x_nranks processes.", "rccl",
                              "rccl", "RCCL over xGMI, one process per GPU (the production path)",
                              "shm", "host-staged sums over POSIX shared memory: ranks of one node that may SHARE a device (bring-up, one-GPU test boxes)",
                              "The reference's only distributed backend owns its communicator in the same place (IpMumpsSolverInterface.cpp:58-75).");
   roptions->AddStringOption1("mi355x_comm_file", "File through which rank 0 hands the RCCL unique id to the other ranks.", "",
                              "*", "any path on a file system all ranks see (default: $MI355X_KKT_COMM_FILE)");
}

void Mi355xSolverInterface::ReadNumericOptions(const OptionsList& options, const std::string& prefix, mi355x_kkt_options& kopts,
      Number& pivtol, Number& pivtolmax)
{
   // options are optional: a host that has not called RegisterOptions simply gets the defaults
   try
   {
      Number v;
      Index iv;
      std::string sv;
      if( options.GetNumericValue("mi355x_pivtol", v, prefix) )
      {
         pivtol = v;
      }
      if( options.GetNumericValue("mi355x_pivtolmax", v, prefix) )
      {
         pivtolmax = v;
      }
      if( options.GetStringValue("mi355x_scaling", sv, prefix) )
      {
         kopts.scaling = (sv == "none") ? 0 : (sv == "matching" ? 4 : (sv == "matching-always" ? 3 : (sv == "matching-device" ? 6 : (sv == "matching-device-always" ? 5 : 1))));
      }
      if( options.GetStringValue("mi355x_ordering", sv, prefix) )
      {
         kopts.ordering = (sv == "md") ? 1 : (sv == "natural" ? 2 : 0);
      }
      if( options.GetStringValue("mi355x_matching", sv, prefix) )
      {
         kopts.matching = (sv == "no") ? 0 : 1;
      }
      if( options.GetIntegerValue("mi355x_nemin", iv, prefix) )
      {
         kopts.nemin = iv;
      }
      if( options.GetIntegerValue("mi355x_nd_leaf", iv, prefix) )
      {
         kopts.nd_leaf = iv;
      }
      if( options.GetIntegerValue("mi355x_max_sn_cols", iv, prefix) )
      {
         kopts.max_sn_cols = iv;
      }
      if( options.GetStringValue("mi355x_smart_quality", sv, prefix) )
      {
         kopts.smart_quality = (sv == "yes") ? 1 : 0;
      }
      if( options.GetIntegerValue("mi355x_delay_rounds", iv, prefix) )
      {
         kopts.delay_rounds = iv;
      }
      if( options.GetIntegerValue("mi355x_device", iv, prefix) )
      {
         kopts.device = iv;
      }
      if( options.GetIntegerValue("mi355x_verbose", iv, prefix) )
      {
         kopts.verbose = iv;
      }
   }
   catch( ... )
   {
      // unregistered options: keep defaults
   }
   if( pivtolmax < pivtol )
   {
      pivtolmax = pivtol;
   }
   kopts.pivtol = pivtol;
   kopts.pivtolmax = pivtolmax;
   kopts.index_base = 1;
}

bool Mi355xSolverInterface::InitializeImpl(const OptionsList& options, const std::string& prefix)
{
   ReadNumericOptions(options, prefix, kopts_, pivtol_, pivtolmax_);
   {
      // mi355x_outer_scaling: the host scales through its TSymScalingMethod hook INSTEAD of the backend -- never both
      std::string osv;
      bool outer = no_internal_scaling_;
      try
      {
         if( options.GetStringValue("mi355x_outer_scaling", osv, prefix) )
         {
            outer = outer || osv != "no";
         }
      }
      catch( ... )
      { }
      if( outer )
      {
         kopts_.scaling = 0;
      }
   }
   if( pivtolmax_ < pivtol_ )
   {
      pivtolmax_ = pivtol_;
   }
   kopts_.pivtol = pivtol_;
   kopts_.pivtolmax = pivtolmax_;
   kopts_.index_base = 1;
   comm_.ReadOptions(options, prefix, kopts_);      // ranks, device, communicator kind (IpMi355xCommBootstrap.hpp)

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
      // new structure: throw away symbolic data (cf. IpMumpsSolverInterface.cpp:227-236)
      if( handle_ )
      {
         mi355x_kkt_destroy(handle_);
         handle_ = NULL;
      }
      if( mi355x_kkt_create(&handle_, &kopts_) != MI355X_KKT_SUCCESS )
      {
         return false;
      }
      analysed_ = false;
      dim_ = 0;
      nonzeros_ = 0;
   }
   else
   {
      mi355x_kkt_set_pivtol(handle_, pivtol_);
      mi355x_kkt_set_pivtolmax(handle_, pivtolmax_);
   }
   pivtol_changed_ = false;
   negevals_ = -1;
   return true;
}

ESymSolverStatus Mi355xSolverInterface::InitializeStructure(Index dim, Index nonzeros, const Index* ia, const Index* ja)
{
   if( warm_start_same_structure_ && analysed_ )
   {
      ASSERT_EXCEPTION(dim_ == dim && nonzeros_ == nonzeros, INVALID_WARMSTART,
                       "Mi355xSolverInterface called with warm_start_same_structure, but the problem size has changed.");
      return SYMSOLVER_SUCCESS;
   }
   dim_ = dim;
   nonzeros_ = nonzeros;
   ia_ = ia;   // owned by TSymLinearSolver, valid for the whole run (IpTSymLinearSolver.cpp:371)
   ja_ = ja;
   analysed_ = false;   // symbolic phase is lazy: it wants the first values for the 2x2 pre-pairing
   staging_.assign(nonzeros > 0 ? nonzeros : 1, 0.);
   return SYMSOLVER_SUCCESS;
}

Number* Mi355xSolverInterface::GetValuesArrayPtr()
{
   if( analysed_ )
   {
      return mi355x_kkt_values_buffer(handle_);
   }
   return &staging_[0];
}

ESymSolverStatus Mi355xSolverInterface::MultiSolve(bool new_matrix, const Index* ia, const Index* ja, Index nrhs,
      Number* rhs_vals, bool check_NegEVals, Index numberOfNegEVals)
{
   (void) ia;
   (void) ja;
   if( !analysed_ )
   {
      if( HaveIpData() )
      {
         IpData().TimingStats().LinearSystemSymbolicFactorization().Start();
      }
      int st = mi355x_kkt_analyse(handle_, dim_, nonzeros_, ia_, ja_, MI355X_KKT_FMT_TRIPLET, &staging_[0]);
      if( HaveIpData() )
      {
         IpData().TimingStats().LinearSystemSymbolicFactorization().End();
      }
      if( st != MI355X_KKT_SUCCESS )
      {
         Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_analyse failed: %s\n", mi355x_kkt_last_error(handle_));
         return SYMSOLVER_FATAL_ERROR;
      }
      Number* buf = mi355x_kkt_values_buffer(handle_);
      if( !buf )
      {
         return SYMSOLVER_FATAL_ERROR;
      }
      std::memcpy(buf, &staging_[0], sizeof(Number) * (size_t) nonzeros_);
      std::vector<Number>().swap(staging_);
      analysed_ = true;
      new_matrix = true;
      if( comm_.Wanted(kopts_) && !comm_.Ready() )
      {
         if( !comm_.Setup(handle_, kopts_, Jnlst()) )
         {
            return SYMSOLVER_FATAL_ERROR;
         }
      }
      mi355x_kkt_info info;
      mi355x_kkt_get_info(handle_, &info);
      Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA,
                     "MI355X analyse: n=%d nnz(A)=%d nnz(L)=%lld flops=%lld supernodes=%d levels=%d maxfront=%d pairs=%d\n",
                     info.n, info.nnz_a, (long long) info.nnz_l, (long long) info.flops_factor, info.num_sn, info.num_levels,
                     info.maxfront, info.num_pairs);
   }

   if( new_matrix || pivtol_changed_ )
   {
      if( HaveIpData() )
      {
         IpData().TimingStats().LinearSystemFactorization().Start();
      }
      int nneg = 0, nzero = 0;
      int st;
      if( new_matrix )
      {
         st = mi355x_kkt_factor(handle_, NULL, &nneg, &nzero);
      }
      else
      {
         st = mi355x_kkt_refactor(handle_, &nneg, &nzero);
      }
      if( HaveIpData() )
      {
         IpData().TimingStats().LinearSystemFactorization().End();
      }
      pivtol_changed_ = false;
      if( st == MI355X_KKT_FATAL )
      {
         Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_factor failed: %s\n", mi355x_kkt_last_error(handle_));
         return SYMSOLVER_FATAL_ERROR;
      }
      negevals_ = nneg;
      Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X factor: %d negative eigenvalues, %d zero pivots (status %d)\n",
                     nneg, nzero, st);
      {
         // what the MA97 adapter prints from info.num_delay (IpMa97SolverInterface.cpp:719-779): delayed pivots since the analysis
         mi355x_kkt_info kinfo;
         if( mi355x_kkt_get_info(handle_, &kinfo) == MI355X_KKT_SUCCESS && (kinfo.num_delayed > 0 || kinfo.num_small > 0) )
         {
            Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X factor: %d delayed pivots in %d structure edits so far, %d pivots forced\n",
                           kinfo.num_delayed, kinfo.num_restructures, kinfo.num_small);
         }
      }
      if( st == MI355X_KKT_SINGULAR )
      {
         return SYMSOLVER_SINGULAR;
      }
      if( check_NegEVals && negevals_ != numberOfNegEVals )
      {
         Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA,
                        "In Mi355xSolverInterface::MultiSolve: wrong inertia: negevals_ = %d, but numberOfNegEVals = %d\n",
                        negevals_, numberOfNegEVals);
         return SYMSOLVER_WRONG_INERTIA;
      }
   }

   if( HaveIpData() )
   {
      IpData().TimingStats().LinearSystemBackSolve().Start();
   }
   int st = mi355x_kkt_solve(handle_, nrhs, rhs_vals, dim_);
   if( HaveIpData() )
   {
      IpData().TimingStats().LinearSystemBackSolve().End();
   }
   if( st != MI355X_KKT_SUCCESS )
   {
      Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_solve failed: %s\n", mi355x_kkt_last_error(handle_));
      return SYMSOLVER_FATAL_ERROR;
   }
   return SYMSOLVER_SUCCESS;
}

ESymSolverStatus Mi355xSolverInterface::DetermineDependentRows(const Index* /*ia*/, const Index* /*ja*/, std::list<Index>& c_deps)
{
   // TSymLinearSolver::DetermineDependentRows (IpTSymLinearSolver.cpp:540-716) has called InitializeStructure for
   // [[I, J^T], [J, 0]] (all diagonal entries present) and filled the values buffer; factor it and report the zero pivots
   c_deps.clear();
   if( !analysed_ )
   {
      if( mi355x_kkt_analyse(handle_, dim_, nonzeros_, ia_, ja_, MI355X_KKT_FMT_TRIPLET, &staging_[0]) != MI355X_KKT_SUCCESS )
      {
         Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_analyse failed: %s\n", mi355x_kkt_last_error(handle_));
         return SYMSOLVER_FATAL_ERROR;
      }
      Number* buf = mi355x_kkt_values_buffer(handle_);
      if( !buf )
      {
         return SYMSOLVER_FATAL_ERROR;
      }
      std::memcpy(buf, &staging_[0], sizeof(Number) * (size_t) nonzeros_);
      std::vector<Number>().swap(staging_);
      analysed_ = true;
   }
   int nneg = 0, nzero = 0;
   // (no equilibration here: exact cancellations between dependent rows should stay exact; the MUMPS adapter likewise
   //  switches its permuting scaling off for this call, IpMumpsSolverInterface.cpp:632-641)
   //  and no delayed-pivot rounds: a dependent row stays a zero pivot wherever it is eliminated, moving it up only costs refactorisations)
   mi355x_kkt_set_scaling(handle_, 0, NULL);
   mi355x_kkt_set_delay_rounds(handle_, 0);
   int st = mi355x_kkt_factor(handle_, NULL, &nneg, &nzero);
   mi355x_kkt_set_delay_rounds(handle_, kopts_.delay_rounds);
   mi355x_kkt_set_scaling(handle_, kopts_.scaling, NULL);
   if( st == MI355X_KKT_FATAL )
   {
      Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_factor failed: %s\n", mi355x_kkt_last_error(handle_));
      return SYMSOLVER_FATAL_ERROR;
   }
   negevals_ = nneg;
   int count = 0;
   std::vector<int> idx(nzero > 0 ? nzero : 1);
   if( mi355x_kkt_zero_pivots(handle_, &idx[0], (int) idx.size(), &count) != MI355X_KKT_SUCCESS )
   {
      return SYMSOLVER_FATAL_ERROR;
   }
   for( int i = 0; i < count && i < (int) idx.size(); ++i )
   {
      c_deps.push_back(idx[i] - 1);      // 0-based, as MUMPS' pivnul_list - 1 (IpMumpsSolverInterface.cpp:703-706)
   }
   return SYMSOLVER_SUCCESS;
}

Index Mi355xSolverInterface::NumberOfNegEVals() const
{
   return negevals_;
}

bool Mi355xSolverInterface::IncreaseQuality()
{
   // same escalation rule as the MA27 / MA97 / SPRAL adapters: u <- min(umax, u^0.75)
   // (IpMa97SolverInterface.cpp:822-854, IpMa27TSolverInterface.cpp:724-740), applied by the library -- which also knows
   // from the last factorisation whether ANY pivot decision depends on u: if none does, a refactorisation would reproduce
   // the same factors, so the honest answer to PDFullSpaceSolver (IpPDFullSpaceSolver.cpp:290-301) is "no".
   double unew = pivtol_;
   if( !handle_ || mi355x_kkt_increase_quality(handle_, &unew) == 0 )
   {
      Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA,
                     "MI355X solver: pivot tolerance %7.2e cannot be increased usefully (maximum reached or no pivot depends on it).\n", pivtol_);
      return false;
   }
   pivtol_changed_ = true;
   Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "Increasing pivot tolerance for MI355X solver from %7.2e to %7.2e.\n", pivtol_, unew);
   pivtol_ = unew;
   return true;
}

SmartPtr<SymLinearSolver> Mi355xAlgorithmBuilder::SymLinearSolverFactory(const Journalist& /*jnlst*/,
      const OptionsList& options, const std::string& prefix)
{
   SmartPtr<SparseSymLinearSolverInterface> iface = new Mi355xSolverInterface();
   SmartPtr<TSymScalingMethod> scaling;
   // `mi355x_outer_scaling yes`: equilibration as a TSymScalingMethod OUTSIDE the backend (computed on the device), under the
   // reference's own control -- with the default linear_scaling_on_demand=yes it only switches on once Ipopt asks for better
   // quality (IpTSymLinearSolver.cpp:429-441), exactly like mc19 does when HSL is linked (IpAlgBuilder.cpp:530-537)
   std::string sv;
   bool outer = false, matching = false;
   try
   {
      if( options.GetStringValue("mi355x_outer_scaling", sv, prefix) )
      {
         outer = sv != "no";
         matching = sv == "matching";
      }
   }
   catch( ... )
   { }
   if( outer )
   {
      // "instead of inside the backend": once linear_scaling_on_demand switches the outer scaling on, the backend must not scale a second time
      static_cast<Mi355xSolverInterface*>(GetRawPtr(iface))->DisableInternalScaling();
      scaling = new Mi355xTSymScalingMethod(matching);
   }
   return new TSymLinearSolver(iface, scaling);
}

} // namespace Ipopt
