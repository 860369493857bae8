// IpMi355xPDSystemSolver.cpp -- see the header.
#include "IpMi355xPDSystemSolver.hpp"
#include "IpPDFullSpaceSolver.hpp"
#include "IpCGPerturbationHandler.hpp"
#include "IpDenseVector.hpp"
#include "IpExpansionMatrix.hpp"
#include "IpIpoptData.hpp"
#include "IpIpoptNLP.hpp"
#include "IpIpoptCalculatedQuantities.hpp"
#include "IpTimingStatistics.hpp"
#include <vector>
#include <cmath>

namespace Ipopt
{

// the primal-dual vectors the handle owns (mi355x_kkt.h: MI355X_KKT_PD_NVEC)
enum { VEC_RHS = 0, VEC_RES = 1, VEC_RESID = 2 };

struct Mi355xPDSystemSolver::Data
{
   SmartPtr<const SymMatrix> W;
   SmartPtr<const Matrix> J_c, J_d, Px_L, Px_U, Pd_L, Pd_U;
   SmartPtr<const Vector> z_L, z_U, v_L, v_U, slack_x_L, slack_x_U, slack_s_L, slack_s_U, sigma_x, sigma_s;
};

static const Number* HostValues(const Vector& v)
{
   const DenseVector* d = dynamic_cast<const DenseVector*>(&v);
   return d ? d->ExpandedValues() : NULL;       // (a homogeneous vector is expanded)
}

Mi355xPDSystemSolver::Mi355xPDSystemSolver(Mi355xAugSystemSolver& aug, PDPerturbationHandler& pert, PDSystemSolver& host_solver)
   : aug_(&aug), pert_(&pert), host_(&host_solver), matrix_cache_(1), augsys_improved_(false), min_refinement_steps_(1),
     max_refinement_steps_(10), residual_ratio_max_(1e-10), residual_ratio_singular_(1e-5), residual_improvement_factor_(0.999999999),
     neg_curv_test_tol_(0.), workspace_asked_(false), n_device_(0), n_host_(0), n_refine_(0)
{
   for( int q = 0; q < 8; ++q )
   {
      data_tags_[q] = 0;
   }
}

Mi355xPDSystemSolver::~Mi355xPDSystemSolver()
{ }

bool Mi355xPDSystemSolver::InitializeImpl(const OptionsList& options, const std::string& prefix)
{
   // the reference's own options (registered by PDFullSpaceSolver::RegisterOptions, IpPDFullSpaceSolver.cpp:42-95)
   options.GetIntegerValue("min_refinement_steps", min_refinement_steps_, prefix);
   options.GetIntegerValue("max_refinement_steps", max_refinement_steps_, prefix);
   options.GetNumericValue("residual_ratio_max", residual_ratio_max_, prefix);
   options.GetNumericValue("residual_ratio_singular", residual_ratio_singular_, prefix);
   options.GetNumericValue("residual_improvement_factor", residual_improvement_factor_, prefix);
   options.GetNumericValue("neg_curv_test_tol", neg_curv_test_tol_, prefix);
   augsys_improved_ = false;
   workspace_asked_ = false;
   for( int q = 0; q < 8; ++q )
   {
      data_tags_[q] = 0;
   }
   // the host implementation initialises the augmented-system solver and the perturbation handler it shares with us
   // (IpPDFullSpaceSolver.cpp:118-126)
   return host_->Initialize(Jnlst(), IpNLP(), IpData(), IpCq(), options, prefix);
}

bool Mi355xPDSystemSolver::DeviceUsable(const IteratesVector& rhs, const IteratesVector& res, const Data& D)
{
   if( neg_curv_test_tol_ > 0. )
   {
      return false;      // inertia-free mode: the curvature test of SolveOnce :592-639 works on host vectors
   }
   const Vector* blocks[16] = {GetRawPtr(rhs.x()), GetRawPtr(rhs.s()), GetRawPtr(rhs.y_c()), GetRawPtr(rhs.y_d()), GetRawPtr(rhs.z_L()),
                               GetRawPtr(rhs.z_U()), GetRawPtr(rhs.v_L()), GetRawPtr(rhs.v_U()), GetRawPtr(res.x()), GetRawPtr(res.s()),
                               GetRawPtr(res.y_c()), GetRawPtr(res.y_d()), GetRawPtr(res.z_L()), GetRawPtr(res.z_U()), GetRawPtr(res.v_L()),
                               GetRawPtr(res.v_U())
                              };
   for( int q = 0; q < 16; ++q )
   {
      if( !dynamic_cast<const DenseVector*>(blocks[q]) )
      {
         return false;
      }
   }
   const Vector* data[8] = {GetRawPtr(D.z_L), GetRawPtr(D.z_U), GetRawPtr(D.v_L), GetRawPtr(D.v_U), GetRawPtr(D.slack_x_L),
                            GetRawPtr(D.slack_x_U), GetRawPtr(D.slack_s_L), GetRawPtr(D.slack_s_U)
                           };
   for( int q = 0; q < 8; ++q )
   {
      if( !dynamic_cast<const DenseVector*>(data[q]) )
      {
         return false;
      }
   }
   const ExpansionMatrix* P[4] = {dynamic_cast<const ExpansionMatrix*>(GetRawPtr(D.Px_L)), dynamic_cast<const ExpansionMatrix*>(GetRawPtr(D.Px_U)),
                                  dynamic_cast<const ExpansionMatrix*>(GetRawPtr(D.Pd_L)), dynamic_cast<const ExpansionMatrix*>(GetRawPtr(D.Pd_U))
                                 };
   if( !P[0] || !P[1] || !P[2] || !P[3] )
   {
      return false;
   }
   if( !workspace_asked_ )
   {
      const Index dims[8] = {rhs.x()->Dim(), rhs.s()->Dim(), rhs.y_c()->Dim(), rhs.y_d()->Dim(), P[0]->NCols(), P[1]->NCols(), P[2]->NCols(),
                             P[3]->NCols()
                            };
      aug_->WantPrimalDualWorkspace(dims, P[0]->ExpandedPosIndices(), P[1]->ExpandedPosIndices(), P[2]->ExpandedPosIndices(),
                                    P[3]->ExpandedPosIndices());
      workspace_asked_ = true;
   }
   return true;
}

bool Mi355xPDSystemSolver::SolveOnceOnDevice(bool treat_as_singular, const Data& D, Index n_cd, int rhs_vec, int res_vec, Number alpha,
      Number beta)
{
   // SolveOnce (IpPDFullSpaceSolver.cpp:377-664) with the reduction / expansion of the bound blocks on the device
   IpData().TimingStats().PDSystemSolverSolveOnce().Start();

   // has the linear system changed since the last call?  (same dependencies as :428-452)
   std::vector<const TaggedObject*> deps(13);
   deps[0] = GetRawPtr(D.W);
   deps[1] = GetRawPtr(D.J_c);
   deps[2] = GetRawPtr(D.J_d);
   deps[3] = GetRawPtr(D.z_L);
   deps[4] = GetRawPtr(D.z_U);
   deps[5] = GetRawPtr(D.v_L);
   deps[6] = GetRawPtr(D.v_U);
   deps[7] = GetRawPtr(D.slack_x_L);
   deps[8] = GetRawPtr(D.slack_x_U);
   deps[9] = GetRawPtr(D.slack_s_L);
   deps[10] = GetRawPtr(D.slack_s_U);
   deps[11] = GetRawPtr(D.sigma_x);
   deps[12] = GetRawPtr(D.sigma_s);
   void* dummy = NULL;
   const bool unchanged = matrix_cache_.GetCachedResult(dummy, deps);
   if( !unchanged )
   {
      matrix_cache_.AddCachedResult(dummy, deps);
      augsys_improved_ = false;
   }

   Number delta_x, delta_s, delta_c, delta_d;
   if( unchanged && !treat_as_singular )
   {
      // same matrix, same perturbations: the factorisation is reused (the augmented-system solver sees no change)
      pert_->CurrentPerturbation(delta_x, delta_s, delta_c, delta_d);
      const ESymSolverStatus st = aug_->Factorize(GetRawPtr(D.W), 1.0, GetRawPtr(D.sigma_x), delta_x, GetRawPtr(D.sigma_s), delta_s,
                                  GetRawPtr(D.J_c), NULL, delta_c, GetRawPtr(D.J_d), NULL, delta_d, false, 0);
      if( st != SYMSOLVER_SUCCESS )
      {
         IpData().TimingStats().PDSystemSolverSolveOnce().End();
         return false;
      }
   }
   else
   {
      // inertia-correction loop (:486-640): the reference's perturbation handler chooses the deltas, we report what the
      // factorisation says about them
      Index count = 0;
      pert_->ConsiderNewSystem(delta_x, delta_s, delta_c, delta_d);
      ESymSolverStatus st = SYMSOLVER_SINGULAR;
      while( st != SYMSOLVER_SUCCESS )
      {
         if( treat_as_singular )
         {
            st = SYMSOLVER_SINGULAR;
            treat_as_singular = false;
         }
         else
         {
            ++count;
            Jnlst().Printf(J_MOREDETAILED, J_LINEAR_ALGEBRA,
                           "MI355X PD: factorising with delta_x=%e delta_s=%e delta_c=%e delta_d=%e\n", delta_x, delta_s, delta_c,
                           delta_d);
            st = aug_->Factorize(GetRawPtr(D.W), 1.0, GetRawPtr(D.sigma_x), delta_x, GetRawPtr(D.sigma_s), delta_s, GetRawPtr(D.J_c), NULL,
                                 delta_c, GetRawPtr(D.J_d), NULL, delta_d, true, n_cd);
         }
         if( st == SYMSOLVER_SUCCESS )
         {
            break;
         }
         bool ok = true;
         if( st == SYMSOLVER_FATAL_ERROR )
         {
            ok = false;
         }
         else if( st == SYMSOLVER_SINGULAR && n_cd > 0 )
         {
            ok = pert_->PerturbForSingularity(delta_x, delta_s, delta_c, delta_d);
            if( !ok )
            {
               Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: no further perturbation for a singular system available.\n");
            }
         }
         else if( st == SYMSOLVER_WRONG_INERTIA && aug_->NumberOfNegEVals() < n_cd )
         {
            // too few negative eigenvalues: numerically singular?  first ask for better pivoting, once per matrix (:541-579)
            Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: fewer negative eigenvalues than constraints.\n");
            bool perturb_as_singular = true;
            if( !augsys_improved_ )
            {
               Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: asking the factorisation for a tighter pivot tolerance.\n");
               augsys_improved_ = aug_->IncreaseQuality();
               if( augsys_improved_ )
               {
                  IpData().Append_info_string("q");
                  perturb_as_singular = false;
               }
               else
               {
                  Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: the pivot tolerance is at its maximum.\n");
               }
            }
            if( perturb_as_singular )
            {
               ok = pert_->PerturbForSingularity(delta_x, delta_s, delta_c, delta_d);
               if( ok )
               {
                  IpData().Append_info_string("a");
               }
               else
               {
                  Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: no further perturbation available (system treated as singular).\n");
               }
            }
         }
         else
         {
            // wrong inertia (too many negative eigenvalues), or singular without constraints
            ok = pert_->PerturbForWrongInertia(delta_x, delta_s, delta_c, delta_d);
            if( !ok )
            {
               Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: no further perturbation for wrong inertia available.\n");
            }
         }
         if( !ok )
         {
            IpData().TimingStats().PDSystemSolverSolveOnce().End();
            return false;
         }
      }
      Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: %" IPOPT_INDEX_FORMAT " trial factorisation(s)\n", count);
      Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA,
                     "MI355X PD: accepted perturbations delta_x=%e delta_s=%e delta_c=%e delta_d=%e\n", delta_x, delta_s, delta_c,
                     delta_d);
      IpData().setPDPert(delta_x, delta_s, delta_c, delta_d);
   }

   // reduce the bound rows into the right-hand side, solve the augmented system, expand (:418-424, :653-661), all on the device
   IpData().TimingStats().LinearSystemBackSolve().Start();
   const int st = mi355x_kkt_pd_solve_once(aug_->Handle(), rhs_vec, res_vec, alpha, beta);
   IpData().TimingStats().LinearSystemBackSolve().End();
   IpData().TimingStats().PDSystemSolverSolveOnce().End();
   if( st != MI355X_KKT_SUCCESS )
   {
      Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_pd_solve_once failed: %s\n", mi355x_kkt_last_error(aug_->Handle()));
      return false;
   }
   return true;
}

bool Mi355xPDSystemSolver::ResidualRatioOnDevice(int rhs_vec, int res_vec, int resid_vec, Number& ratio)
{
   // ComputeResiduals + ComputeResidualRatio (:666-820)
   IpData().TimingStats().ComputeResiduals().Start();
   Number deltas[4];
   pert_->CurrentPerturbation(deltas[0], deltas[1], deltas[2], deltas[3]);
   double norms[3] = {0., 0., 0.};
   const int st = mi355x_kkt_pd_residual(aug_->Handle(), rhs_vec, res_vec, resid_vec, deltas, norms);
   IpData().TimingStats().ComputeResiduals().End();
   if( st != MI355X_KKT_SUCCESS )
   {
      Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "mi355x_kkt_pd_residual failed: %s\n", mi355x_kkt_last_error(aug_->Handle()));
      return false;
   }
   const Number nrm_rhs = norms[0], nrm_res = norms[1], nrm_resid = norms[2];
   Jnlst().Printf(J_MOREDETAILED, J_LINEAR_ALGEBRA, "MI355X PD: max norms rhs %8.2e sol %8.2e resid %8.2e\n", nrm_rhs, nrm_res, nrm_resid);
   if( nrm_rhs + nrm_res == 0. )
   {
      ratio = nrm_resid;
   }
   else
   {
      const Number max_cond = 1e6;
      ratio = nrm_resid / (Min(nrm_res, max_cond * nrm_rhs) + nrm_rhs);
   }
   Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: residual ratio %e\n", ratio);
   return true;
}

bool Mi355xPDSystemSolver::Solve(Number alpha, Number beta, const IteratesVector& rhs, IteratesVector& res, bool allow_inexact,
                                 bool improve_solution)
{
   // the data of the current iterate (:171-191)
   Data D;
   D.W = IpData().W();
   D.J_c = IpCq().curr_jac_c();
   D.J_d = IpCq().curr_jac_d();
   D.Px_L = IpNLP().Px_L();
   D.Px_U = IpNLP().Px_U();
   D.Pd_L = IpNLP().Pd_L();
   D.Pd_U = IpNLP().Pd_U();
   D.z_L = IpData().curr()->z_L();
   D.z_U = IpData().curr()->z_U();
   D.v_L = IpData().curr()->v_L();
   D.v_U = IpData().curr()->v_U();
   D.slack_x_L = IpCq().curr_slack_x_L();
   D.slack_x_U = IpCq().curr_slack_x_U();
   D.slack_s_L = IpCq().curr_slack_s_L();
   D.slack_s_U = IpCq().curr_slack_s_U();
   D.sigma_x = IpCq().curr_sigma_x();
   D.sigma_s = IpCq().curr_sigma_s();

   if( !DeviceUsable(rhs, res, D) )
   {
      ++n_host_;
      return host_->Solve(alpha, beta, rhs, res, allow_inexact, improve_solution);
   }

   IpData().TimingStats().PDSystemSolverTotal().Start();
   mi355x_kkt_handle h = aug_->Handle();
   const Index n_cd = rhs.y_c()->Dim() + rhs.y_d()->Dim();

   SmartPtr<IteratesVector> copy_res;
   if( beta != 0. )
   {
      copy_res = res.MakeNewIteratesVectorCopy();
   }

   bool done = false;
   bool refactor_with_new_pivtol = false;   // the pivot tolerance was raised: factor and solve again
   bool treat_as_singular = false;              // refinement failed: see whether a perturbed system does better
   bool singular_already_tried = false;
   bool uploaded = false;
   bool ok = true;

   while( !done && ok )
   {
      if( !aug_->HasPrimalDualWorkspace() )
      {
         // the analysis has not happened yet (it normally has: the least-square multipliers come first): a factor-only call with
         // the current perturbations creates analysis and workspace without changing what SolveOnce will decide
         Number dx, ds, dc, dd;
         pert_->CurrentPerturbation(dx, ds, dc, dd);
         const ESymSolverStatus st0 = aug_->Factorize(GetRawPtr(D.W), 1.0, GetRawPtr(D.sigma_x), dx, GetRawPtr(D.sigma_s), ds, GetRawPtr(D.J_c),
                                      NULL, dc, GetRawPtr(D.J_d), NULL, dd, false, 0);
         aug_->ForgetFactorization();        // (SolveOnce asks its own inertia question)
         if( st0 == SYMSOLVER_FATAL_ERROR || !aug_->HasPrimalDualWorkspace() )
         {
            ok = false;
            break;
         }
      }
      if( !uploaded )
      {
         // bound multipliers and slacks of the iterate, when they changed
         const Vector* dat[8] = {GetRawPtr(D.z_L), GetRawPtr(D.z_U), GetRawPtr(D.v_L), GetRawPtr(D.v_U), GetRawPtr(D.slack_x_L),
                                 GetRawPtr(D.slack_x_U), GetRawPtr(D.slack_s_L), GetRawPtr(D.slack_s_U)
                                };
         bool changed = false;
         for( int q = 0; q < 8; ++q )
         {
            if( dat[q]->GetTag() != data_tags_[q] )
            {
               changed = true;
            }
         }
         if( changed )
         {
            const Number* arr[8];
            for( int q = 0; q < 8; ++q )
            {
               arr[q] = HostValues(*dat[q]);
               data_tags_[q] = dat[q]->GetTag();
            }
            ok = mi355x_kkt_pd_put_data(h, arr) == MI355X_KKT_SUCCESS;
         }
         const Number* rb[8] = {HostValues(*rhs.x()), HostValues(*rhs.s()), HostValues(*rhs.y_c()), HostValues(*rhs.y_d()), HostValues(*rhs.z_L()),
                                HostValues(*rhs.z_U()), HostValues(*rhs.v_L()), HostValues(*rhs.v_U())
                               };
         ok = ok && mi355x_kkt_pd_put(h, VEC_RHS, rb) == MI355X_KKT_SUCCESS;
         if( improve_solution )
         {
            const Number* sb[8] = {HostValues(*res.x()), HostValues(*res.s()), HostValues(*res.y_c()), HostValues(*res.y_d()), HostValues(*res.z_L()),
                                   HostValues(*res.z_U()), HostValues(*res.v_L()), HostValues(*res.v_U())
                                  };
            ok = ok && mi355x_kkt_pd_put(h, VEC_RES, sb) == MI355X_KKT_SUCCESS;
         }
         uploaded = true;
         if( !ok )
         {
            break;
         }
      }

      // with improve_solution the caller hands over a solution: the first solve is skipped (:213-222)
      bool solved = true;
      if( !improve_solution )
      {
         solved = SolveOnceOnDevice(treat_as_singular, D, n_cd, VEC_RHS, VEC_RES, 1., 0.);
         refactor_with_new_pivtol = false;
         treat_as_singular = false;
      }
      else
      {
         // no solve precedes the first residual: bring the device-side sources (W, J_c, J_d, Sigma) and the factorisation up to the CURRENT
         // iterate first -- what was assembled last may belong to an earlier iterate or to a least-square / restoration call (W_factor = 0);
         // nothing is uploaded or factored if nothing has changed since
         Number dx, ds, dc, dd;
         pert_->CurrentPerturbation(dx, ds, dc, dd);
         solved = aug_->Factorize(GetRawPtr(D.W), 1.0, GetRawPtr(D.sigma_x), dx, GetRawPtr(D.sigma_s), ds, GetRawPtr(D.J_c), NULL, dc,
                                  GetRawPtr(D.J_d), NULL, dd, false, 0) == SYMSOLVER_SUCCESS;
      }
      improve_solution = false;
      if( !solved )
      {
         // not solvable as it stands: the caller deals with it (:224-230)
         IpData().TimingStats().PDSystemSolverTotal().End();
         return false;
      }
      if( allow_inexact )
      {
         break;      // no safety net asked for
      }

      Number residual_ratio = 0.;
      if( !ResidualRatioOnDevice(VEC_RHS, VEC_RES, VEC_RESID, residual_ratio) )
      {
         ok = false;
         break;
      }
      Number ratio_before = residual_ratio;

      // iterative refinement on the unreduced system (:262-346)
      Index nsteps = 0;
      bool give_up = false;
      while( !give_up && (nsteps < min_refinement_steps_ || residual_ratio > residual_ratio_max_) )
      {
         // res <- res - K8^{-1} resid
         if( !SolveOnceOnDevice(false, D, n_cd, VEC_RESID, VEC_RES, -1., 1.) )
         {
            IpData().TimingStats().PDSystemSolverTotal().End();
            THROW_EXCEPTION(INTERNAL_ABORT, "SolveOnce returns false during iterative refinement.");
         }
         if( !ResidualRatioOnDevice(VEC_RHS, VEC_RES, VEC_RESID, residual_ratio) )
         {
            ok = false;
            break;
         }
         ++nsteps;
         ++n_refine_;
         // give up? (:285-342)
         if( residual_ratio > residual_ratio_max_ && nsteps > min_refinement_steps_
             && (nsteps > max_refinement_steps_ || residual_ratio > residual_improvement_factor_ * ratio_before) )
         {
            Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: refinement stalled at residual ratio %e\n", residual_ratio);
            give_up = true;
            refactor_with_new_pivtol = false;
            if( !singular_already_tried )
            {
               // first a better factorisation (once per linear system), then "the modification is singular" -- and that only when
               // the residual is really bad
               if( !augsys_improved_ )
               {
                  Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: asking the factorisation for a tighter pivot tolerance.\n");
                  augsys_improved_ = aug_->IncreaseQuality();
                  if( augsys_improved_ )
                  {
                     IpData().Append_info_string("q");
                     refactor_with_new_pivtol = true;
                  }
                  else
                  {
                     treat_as_singular = true;
                  }
               }
               else
               {
                  treat_as_singular = true;
               }
               singular_already_tried = treat_as_singular;
               if( treat_as_singular )
               {
                  if( residual_ratio < residual_ratio_singular_ )
                  {
                     treat_as_singular = false;
                     IpData().Append_info_string("S");
                     Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: residual small enough, solution kept.\n");
                  }
                  else
                  {
                     IpData().Append_info_string("s");
                     Jnlst().Printf(J_DETAILED, J_LINEAR_ALGEBRA, "MI355X PD: perturbing the system as if it were singular.\n");
                  }
               }
            }
            else
            {
               treat_as_singular = false;
            }
         }
         ratio_before = residual_ratio;
      }
      done = !refactor_with_new_pivtol && !treat_as_singular;
   }

   if( !ok )
   {
      IpData().TimingStats().PDSystemSolverTotal().End();
      Jnlst().Printf(J_ERROR, J_LINEAR_ALGEBRA, "MI355X primal-dual workspace failed: %s\n", mi355x_kkt_last_error(h));
      THROW_EXCEPTION(INTERNAL_ABORT, "Mi355xPDSystemSolver: device operation failed.");
   }

   // the result comes down once; alpha and beta are applied as the reference applies them (:349-358)
   Number* ob[8] = {static_cast<DenseVector*>(GetRawPtr(res.x_NonConst()))->Values(), static_cast<DenseVector*>(GetRawPtr(res.s_NonConst()))->Values(),
                    static_cast<DenseVector*>(GetRawPtr(res.y_c_NonConst()))->Values(), static_cast<DenseVector*>(GetRawPtr(res.y_d_NonConst()))->Values(),
                    static_cast<DenseVector*>(GetRawPtr(res.z_L_NonConst()))->Values(), static_cast<DenseVector*>(GetRawPtr(res.z_U_NonConst()))->Values(),
                    static_cast<DenseVector*>(GetRawPtr(res.v_L_NonConst()))->Values(), static_cast<DenseVector*>(GetRawPtr(res.v_U_NonConst()))->Values()
                   };
   if( mi355x_kkt_pd_get(h, VEC_RES, ob) != MI355X_KKT_SUCCESS )
   {
      IpData().TimingStats().PDSystemSolverTotal().End();
      THROW_EXCEPTION(INTERNAL_ABORT, "Mi355xPDSystemSolver: download of the solution failed.");
   }
   if( alpha != 0. )
   {
      res.Scal(alpha);
   }
   if( beta != 0. )
   {
      res.Axpy(beta, *copy_res);
   }
   ++n_device_;
   IpData().TimingStats().PDSystemSolverTotal().End();
   return true;
}

// ---- builder -----------------------------------------------------------------------------------------------------------------

namespace
{
class Mi355xPDAlgorithmBuilder: public AlgorithmBuilder
{
public:
   Mi355xPDAlgorithmBuilder(const SmartPtr<Mi355xAugSystemSolver>& aug)
      : AlgorithmBuilder(GetRawPtr(aug), "mi355x-ldlt (device-side KKT assembly and primal-dual refinement)"), aug_(aug)
   { }

   virtual SmartPtr<PDSystemSolver> PDSystemSolverFactory(const Journalist& jnlst, const OptionsList& options, const std::string& prefix)
   {
      // as the reference (IpAlgBuilder.cpp:644-664) ...
      SmartPtr<PDPerturbationHandler> pert;
      std::string lsmethod;
      options.GetStringValue("line_search_method", lsmethod, prefix);
      if( lsmethod == "cg-penalty" )
      {
         pert = new CGPerturbationHandler();
      }
      else
      {
         pert = new PDPerturbationHandler();
      }
      SmartPtr<AugSystemSolver> top = GetAugSystemSolver(jnlst, options, prefix);
      SmartPtr<PDSystemSolver> host = new PDFullSpaceSolver(*top, *pert);
      // ... with the device implementation in front of it when the augmented-system solver IS ours (a limited-memory Hessian
      // wraps it into a low-rank solver: that route stays on the host)
      if( GetRawPtr(top) != static_cast<AugSystemSolver*>(GetRawPtr(aug_)) )
      {
         return host;
      }
      created_ = new Mi355xPDSystemSolver(*aug_, *pert, *host);
      return GetRawPtr(created_);
   }

   SmartPtr<Mi355xPDSystemSolver> created_;

private:
   SmartPtr<Mi355xAugSystemSolver> aug_;
};
}

bool GetMi355xPDSystemStatistics(const SmartPtr<AlgorithmBuilder>& builder, Index& device_solves, Index& host_solves, Index& refinement_steps)
{
   const Mi355xPDAlgorithmBuilder* b = dynamic_cast<const Mi355xPDAlgorithmBuilder*>(GetRawPtr(builder));
   if( !b || IsNull(b->created_) )
   {
      return false;
   }
   device_solves = b->created_->DeviceSolves();
   host_solves = b->created_->HostSolves();
   refinement_steps = b->created_->RefinementSteps();
   return true;
}

SmartPtr<AlgorithmBuilder> MakeMi355xPDSystemAlgorithmBuilder()
{
   SmartPtr<Mi355xAugSystemSolver> aug = new Mi355xAugSystemSolver();
   return new Mi355xPDAlgorithmBuilder(aug);
}

} // namespace Ipopt
