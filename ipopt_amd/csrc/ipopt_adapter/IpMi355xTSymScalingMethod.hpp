// IpMi355xTSymScalingMethod.hpp -- a TSymScalingMethod (reference
// src/Algorithm/LinearSolvers/IpTSymScalingMethod.hpp:20-60) that computes symmetric Ruiz inf-norm equilibration factors
// of the triplet KKT matrix ON THE DEVICE (mi355x_kkt_ruiz_scaling).  It is the counterpart of Mc19TSymScalingMethod
// (IpMc19TSymScalingMethod.cpp:100-204) for hosts without HSL: plugged into TSymLinearSolver it makes the reference's
// `linear_scaling_on_demand` logic (IpTSymLinearSolver.cpp:429-441: scaling switched on by the first IncreaseQuality
// request) work with this backend unchanged.
#ifndef IPMI355XTSYMSCALINGMETHOD_HPP
#define IPMI355XTSYMSCALINGMETHOD_HPP

#include "IpTSymScalingMethod.hpp"
#include "mi355x_kkt.h"

namespace Ipopt
{

class Mi355xTSymScalingMethod: public TSymScalingMethod
{
public:
   /** matching = false: Ruiz equilibration on the device; true: maximum-product matching scaling (MC64-style, host) */
   Mi355xTSymScalingMethod(bool matching = false, int device = -1, int sweeps = 4)
      : matching_(matching), device_(device), sweeps_(sweeps)
   { }
   virtual ~Mi355xTSymScalingMethod()
   { }
   virtual bool InitializeImpl(const OptionsList& /*options*/, const std::string& /*prefix*/)
   {
      return true;
   }
   virtual bool ComputeSymTScalingFactors(Index n, Index nnz, const Index* airn, const Index* ajcn, const Number* a,
                                          Number* scaling_factors)
   {
      if( matching_ )
      {
         return mi355x_kkt_matching_scaling(n, nnz, airn, ajcn, a, 1, scaling_factors, NULL) == MI355X_KKT_SUCCESS;
      }
      return mi355x_kkt_ruiz_scaling(device_, n, nnz, airn, ajcn, a, 1, sweeps_, scaling_factors) == MI355X_KKT_SUCCESS;
   }
private:
   Mi355xTSymScalingMethod(const Mi355xTSymScalingMethod&);
   void operator=(const Mi355xTSymScalingMethod&);
   bool matching_;
   int device_, sweeps_;
};

} // namespace Ipopt
#endif
