// IpMi355xPDSystemSolver.hpp -- Ipopt plug-in for SURVEY 8(f)2: a PDSystemSolver (reference
// src/Algorithm/IpPDSystemSolver.hpp:80-150) for the MI355X backend that keeps the primal-dual vectors ON THE DEVICE during a
// Solve.  The reference's PDFullSpaceSolver (IpPDFullSpaceSolver.cpp:128-375) runs, per Solve, one augmented-system solve plus
// one per refinement step, each a host -> device -> host trip of the right-hand side, and between them the residual of the
// unreduced 8-block system (:666-793) as single-threaded host sparse products: at n = 10^6 that host work is 3/4 of the
// PDSystemSolverTotal timer once the factorisation runs on the GPU.  Here the right-hand side goes up once, reduce / solve /
// expand / residual / norms run as kernels (mi355x_kkt_pd_*), the result comes down once.
//
// What stays exactly the reference's: the control flow of Solve (refinement until residual_ratio <= residual_ratio_max,
// min / max_refinement_steps, residual_improvement_factor, IncreaseQuality, "pretend singular"), the inertia-correction loop of
// SolveOnce with the reference's own PDPerturbationHandler object (delta heuristics, IpPDPerturbationHandler.cpp:144-452),
// the options, the timers (PDSystemSolverTotal, PDSystemSolverSolveOnce, ComputeResiduals) and the info-string characters.
// Anything the device path does not cover (vectors that are not DenseVectors, bound matrices that are not ExpansionMatrices,
// the inertia-free curvature test neg_curv_test_tol > 0, a wrapped augmented-system solver) is handed to a reference
// PDFullSpaceSolver built on the same AugSystemSolver and perturbation handler.
//
// Needs IpPDPerturbationHandler.hpp / IpPDFullSpaceSolver.hpp, which Ipopt does not install: this class is built inside the
// Ipopt tree (the B1' patch route) or against the source headers (oracle/ref_build.mk).
#ifndef IPMI355XPDSYSTEMSOLVER_HPP
#define IPMI355XPDSYSTEMSOLVER_HPP

#include "IpPDSystemSolver.hpp"
#include "IpPDPerturbationHandler.hpp"
#include "IpCachedResults.hpp"
#include "IpAlgBuilder.hpp"
#include "IpMi355xAugSystemSolver.hpp"

namespace Ipopt
{

class Mi355xPDSystemSolver: public PDSystemSolver
{
public:
   /** aug must be the solver the algorithm uses for the augmented system; host_solver the reference implementation on the same
    *  objects (taken for everything the device path does not cover) */
   Mi355xPDSystemSolver(Mi355xAugSystemSolver& aug, PDPerturbationHandler& pert, PDSystemSolver& host_solver);
   virtual ~Mi355xPDSystemSolver();

   bool InitializeImpl(const OptionsList& options, const std::string& prefix);

   virtual bool Solve(Number alpha, Number beta, const IteratesVector& rhs, IteratesVector& res, bool allow_inexact = false,
                      bool improve_solution = false);

   /** Solve calls answered on the device / handed to the host implementation (tests, logging) */
   Index DeviceSolves() const
   {
      return n_device_;
   }
   Index HostSolves() const
   {
      return n_host_;
   }
   Index RefinementSteps() const
   {
      return n_refine_;
   }

private:
   Mi355xPDSystemSolver();
   Mi355xPDSystemSolver(const Mi355xPDSystemSolver&);
   void operator=(const Mi355xPDSystemSolver&);

   struct Data;      // the matrices and vectors of the current iterate (what PDFullSpaceSolver::Solve gathers at :171-191)

   bool DeviceUsable(const IteratesVector& rhs, const IteratesVector& res, const Data& D);
   bool SolveOnceOnDevice(bool pretend_singular, const Data& D, Index n_cd, int rhs_vec, int res_vec, Number alpha, Number beta);
   bool ResidualRatioOnDevice(int rhs_vec, int res_vec, int resid_vec, Number& ratio);

   SmartPtr<Mi355xAugSystemSolver> aug_;
   SmartPtr<PDPerturbationHandler> pert_;
   SmartPtr<PDSystemSolver> host_;
   CachedResults<void*> matrix_cache_;          // "has anything of the linear system changed": the test of SolveOnce :428-452
   bool augsys_improved_;
   Index min_refinement_steps_, max_refinement_steps_;
   Number residual_ratio_max_, residual_ratio_singular_, residual_improvement_factor_, neg_curv_test_tol_;
   bool workspace_asked_;
   TaggedObject::Tag data_tags_[8];             // what the device copy of z_L .. slack_s_U was uploaded from
   Index n_device_, n_host_, n_refine_;
};

/** AlgorithmBuilder for the full device route: Mi355xAugSystemSolver as the custom augmented-system solver (`linear_solver`
 *  must be "custom", IpAlgBuilder.cpp:576-584) and Mi355xPDSystemSolver through the virtual PDSystemSolverFactory
 *  (IpAlgBuilder.hpp:138). */
SmartPtr<AlgorithmBuilder> MakeMi355xPDSystemAlgorithmBuilder();
/** counters of the Mi355xPDSystemSolver such a builder created (false: it has not created one); tests, logging */
bool GetMi355xPDSystemStatistics(const SmartPtr<AlgorithmBuilder>& builder, Index& device_solves, Index& host_solves, Index& refinement_steps);

} // namespace Ipopt
#endif
