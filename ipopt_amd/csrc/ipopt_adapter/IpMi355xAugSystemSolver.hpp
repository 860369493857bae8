// IpMi355xAugSystemSolver.hpp -- Ipopt plug-in, route (ii) of SURVEY 8(b): a custom AugSystemSolver
// (reference src/Algorithm/IpAugSystemSolver.hpp:40-230) that replaces, for the MI355X backend, the chain
//     StdAugSystemSolver (IpStdAugSystemSolver.cpp:81-230,309-468: CompoundSymMatrix object graph)
//  -> TSymLinearSolver   (IpTSymLinearSolver.cpp:159-312,453-533: TripletHelper::FillValues over the whole matrix)
//  -> SparseSymLinearSolverInterface (host value buffer, 8 nnz bytes over PCIe per factorisation)
// by DEVICE-SIDE assembly of the KKT values: the pieces Ipopt hands over -- W, J_c, J_d values, the
// diagonals D_x, D_s, D_c, D_d and the four perturbations delta_* -- are uploaded one by one, only when THEIR tag changed,
// and mi355x_kkt_factor_assembled forms  W_factor W + D_x + delta_x I,  D_s + delta_s I,  J_c,  D_c - delta_c I,  J_d,  -I,
// D_d - delta_d I  on the GPU.  An inertia-correction retry (IpPDFullSpaceSolver.cpp:486-640) uploads nothing.
//
// The triplet structure is laid out exactly as TripletHelper::FillRowCol_(CompoundSymMatrix) would lay it out
// (IpTripletHelper.cpp:805-842 over the block order of IpStdAugSystemSolver.cpp:263-298), so the symbolic analysis -- and
// with it every pivot, inertia and iterate -- is identical to the route through Mi355xSolverInterface.
#ifndef IPMI355XAUGSYSTEMSOLVER_HPP
#define IPMI355XAUGSYSTEMSOLVER_HPP

#include "IpAugSystemSolver.hpp"
#include "IpAlgBuilder.hpp"
#include "mi355x_kkt.h"
#include "IpMi355xCommBootstrap.hpp"
#include <vector>
#include <string>

namespace Ipopt
{

class Mi355xAugSystemSolver: public AugSystemSolver
{
public:
   Mi355xAugSystemSolver();
   virtual ~Mi355xAugSystemSolver();

   bool InitializeImpl(const OptionsList& options, const std::string& prefix);

   ESymSolverStatus MultiSolve(const SymMatrix* W, Number W_factor, const Vector* D_x, Number delta_x, const Vector* D_s,
                               Number delta_s, const Matrix* J_c, const Vector* D_c, Number delta_c, const Matrix* J_d,
                               const Vector* D_d, Number delta_d, std::vector<SmartPtr<const Vector> >& rhs_xV,
                               std::vector<SmartPtr<const Vector> >& rhs_sV, std::vector<SmartPtr<const Vector> >& rhs_cV,
                               std::vector<SmartPtr<const Vector> >& rhs_dV, std::vector<SmartPtr<Vector> >& sol_xV,
                               std::vector<SmartPtr<Vector> >& sol_sV, std::vector<SmartPtr<Vector> >& sol_cV,
                               std::vector<SmartPtr<Vector> >& sol_dV, bool check_NegEVals, Index numberOfNegEVals);

   Index NumberOfNegEVals() const
   {
      return negevals_;
   }
   bool ProvidesInertia() const
   {
      return true;
   }
   bool IncreaseQuality();

   /** @name what Mi355xPDSystemSolver needs besides the AugSystemSolver contract */
   ///@{
   /** bring the factorisation of the given augmented system up to date WITHOUT solving: same change test, uploads, statuses
    *  (SINGULAR / WRONG_INERTIA) as MultiSolve */
   ESymSolverStatus Factorize(const SymMatrix* W, Number W_factor, const Vector* D_x, Number delta_x, const Vector* D_s, Number delta_s,
                              const Matrix* J_c, const Vector* D_c, Number delta_c, const Matrix* J_d, const Vector* D_d, Number delta_d,
                              bool check_NegEVals, Index numberOfNegEVals);
   /** ask for the primal-dual device workspace (mi355x_kkt_pd_define) to be created with the analysis; dims = {n_x, n_s, n_c, n_d,
    *  n_xL, n_xU, n_sL, n_sU}, idx_* = ExpansionMatrix::ExpandedPosIndices of Px_L, Px_U, Pd_L, Pd_U.  Call before the first Factorize. */
   void WantPrimalDualWorkspace(const Index dims[8], const Index* idx_xl, const Index* idx_xu, const Index* idx_sl, const Index* idx_su);
   bool HasPrimalDualWorkspace() const
   {
      return pd_defined_;
   }
   /** the next Factorize / MultiSolve factors again (and answers the inertia question again) even if nothing changed */
   void ForgetFactorization()
   {
      have_factor_ = false;
   }
   mi355x_kkt_handle Handle() const
   {
      return handle_;
   }
   ///@}

   /** bytes uploaded for matrix values so far / number of factorisations that uploaded nothing (tests, logging) */
   long long UploadedBytes() const
   {
      return uploaded_bytes_;
   }
   Index FactorizationsWithoutUpload() const
   {
      return nfact_noupload_;
   }

private:
   Mi355xAugSystemSolver(const Mi355xAugSystemSolver&);
   void operator=(const Mi355xAugSystemSolver&);

   enum Segment { SEG_W = 0, SEG_DX, SEG_DS, SEG_JC, SEG_DC, SEG_JD, SEG_ID, SEG_DD, NSEG };

   void BuildStructure(const SymMatrix& W, const Matrix& J_c, const Matrix& J_d);
   bool DefinePrimalDualWorkspace();
   ESymSolverStatus EnsureFactorization(const SymMatrix* W, Number W_factor, const Vector* D_x, Number delta_x, const Vector* D_s,
                                        Number delta_s, const Matrix* J_c, const Vector* D_c, Number delta_c, const Matrix* J_d,
                                        const Vector* D_d, Number delta_d, bool check_NegEVals, Index numberOfNegEVals);
   /** refresh the segments whose source object changed; returns true if the matrix differs from the factored one */
   bool UpdateSources(const SymMatrix* W, Number W_factor, const Vector* D_x, Number delta_x, const Vector* D_s, Number delta_s,
                      const Matrix& J_c, const Vector* D_c, Number delta_c, const Matrix& J_d, const Vector* D_d, Number delta_d,
                      bool upload);

   mi355x_kkt_handle handle_;
   mi355x_kkt_options kopts_;
   Mi355xCommBootstrap comm_;          // ranks sharing the factorisation (mi355x_nranks / mi355x_rank / mi355x_comm)
   bool structured_, analysed_, have_factor_, pivtol_changed_, warm_start_same_structure_;
   Number pivtol_, pivtolmax_;
   Index negevals_;
   Index n_x_, n_s_, n_c_, n_d_, dim_, nnz_;
   Index nnz_w_, nnz_jc_, nnz_jd_;
   std::vector<Index> irn_, jcn_;
   long long seg_off_[NSEG], seg_len_[NSEG];
   double scale_[NSEG], shift_[NSEG];
   // what the device sources currently hold (tags of the objects they were filled from; 0 = none)
   TaggedObject::Tag w_tag_, jc_tag_, jd_tag_, dx_tag_, ds_tag_, dc_tag_, dd_tag_;
   std::vector<Number> first_vals_;    // host copy of the sources before the (lazy) analysis has created the device buffers
   long long uploaded_bytes_;
   Index nfact_noupload_;
   bool pd_wanted_, pd_defined_, singular_;
   int32_t pd_dims_[8];
   std::vector<int32_t> pd_idx_[4];
};

/** AlgorithmBuilder that installs the custom AugSystemSolver through the reference's own constructor argument
 *  (IpAlgBuilder.hpp:55-58, IpAlgBuilder.cpp:82-88,576-584; `linear_solver` must be "custom"). */
SmartPtr<AlgorithmBuilder> MakeMi355xAugSystemAlgorithmBuilder();

} // namespace Ipopt
#endif
