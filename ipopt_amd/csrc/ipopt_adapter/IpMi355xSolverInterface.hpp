// IpMi355xSolverInterface.hpp -- Ipopt plug-in (route B1 of SURVEY 8(b)): an implementation of the
// reference's abstract SparseSymLinearSolverInterface
// (reference src/Algorithm/LinearSolvers/IpSparseSymLinearSolverInterface.hpp:98-256) that forwards to
// the C ABI of include/mi355x_kkt.h.  Compiled against the reference's installed headers only; the
// rest of src/Algorithm runs unmodified on top of it.
//
// Behavioural template: the reference's MA97 / SPRAL adapters (IpMa97SolverInterface.cpp:611-820,
// IpSpralSolverInterface.cpp:510-708) -- keep a private copy of the values so that a pivot-tolerance
// change can re-factor without SYMSOLVER_CALL_AGAIN (SURVEY 8(b) pitfall 7) -- and the MUMPS adapter's
// lazy symbolic phase (IpMumpsSolverInterface.cpp:349-383).
#ifndef IPMI355XSOLVERINTERFACE_HPP
#define IPMI355XSOLVERINTERFACE_HPP

#include "IpSparseSymLinearSolverInterface.hpp"
#include "IpAlgBuilder.hpp"
#include "IpRegOptions.hpp"
#include "mi355x_kkt.h"
#include "IpMi355xCommBootstrap.hpp"
#include <vector>
#include <list>
#include <string>

namespace Ipopt
{

class Mi355xSolverInterface: public SparseSymLinearSolverInterface
{
public:
   Mi355xSolverInterface();
   virtual ~Mi355xSolverInterface();

   bool InitializeImpl(const OptionsList& options, const std::string& prefix);

   ESymSolverStatus InitializeStructure(Index dim, Index nonzeros, const Index* ia, const Index* ja);
   Number* GetValuesArrayPtr();
   ESymSolverStatus MultiSolve(bool new_matrix, const Index* ia, const Index* ja, Index nrhs, Number* rhs_vals,
                               bool check_NegEVals, Index numberOfNegEVals);
   Index NumberOfNegEVals() const;
   bool IncreaseQuality();
   /** The host scales OUTSIDE the backend (mi355x_outer_scaling: Mi355xTSymScalingMethod behind linear_scaling_on_demand): the backend's
    *  own equilibration is switched off for good, whatever mi355x_scaling says -- otherwise the matrix would be scaled twice. */
   void DisableInternalScaling()
   {
      no_internal_scaling_ = true;
   }
   bool ProvidesInertia() const
   {
      return true;
   }
   EMatrixFormat MatrixFormat() const
   {
      return Triplet_Format;   // duplicates / mixed triangles are canonicalised by our own analysis
   }

   /** degeneracy detection (IpSparseSymLinearSolverInterface.hpp:240-255): the zero pivots of the factorisation of
    *  [[I, J^T], [J, 0]] are the linearly dependent rows of J (what the MUMPS adapter does, IpMumpsSolverInterface.cpp:617-709) */
   bool ProvidesDegeneracyDetection() const
   {
      return true;
   }
   ESymSolverStatus DetermineDependentRows(const Index* ia, const Index* ja, std::list<Index>& c_deps);

   static void RegisterOptions(SmartPtr<RegisteredOptions> roptions);
   /** the numeric mi355x_* options into a C-ABI option block (shared with Mi355xAugSystemSolver) */
   static void ReadNumericOptions(const OptionsList& options, const std::string& prefix, mi355x_kkt_options& kopts, Number& pivtol,
                                  Number& pivtolmax);

private:
   Mi355xSolverInterface(const Mi355xSolverInterface&);
   void operator=(const Mi355xSolverInterface&);

   mi355x_kkt_handle handle_;
   mi355x_kkt_options kopts_;
   Index dim_, nonzeros_;
   const Index* ia_;
   const Index* ja_;
   bool analysed_;
   bool pivtol_changed_;
   bool warm_start_same_structure_;
   Number pivtol_, pivtolmax_;
   Index negevals_;
   std::vector<Number> staging_;   // values before the (lazy) analysis has produced the pinned buffer
   // multi-GPU (options mi355x_nranks / mi355x_rank / mi355x_comm / mi355x_comm_file, or the launcher's environment)
   Mi355xCommBootstrap comm_;
   bool no_internal_scaling_ = false;
};

/** AlgorithmBuilder that injects the MI355X backend through the reference's own virtual factory
 *  (IpAlgBuilder.hpp:88; selection route (i) of SURVEY 8(b)). */
class Mi355xAlgorithmBuilder: public AlgorithmBuilder
{
public:
   Mi355xAlgorithmBuilder()
      : AlgorithmBuilder()
   { }
   virtual SmartPtr<SymLinearSolver> SymLinearSolverFactory(const Journalist& jnlst, const OptionsList& options,
         const std::string& prefix);
};

} // namespace Ipopt
#endif
