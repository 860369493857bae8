// IpMi355xCommBootstrap.hpp -- the multi-rank side of the MI355X plug-ins, shared by Mi355xSolverInterface (route B1) and
// Mi355xAugSystemSolver (routes ii / iii): who am I among the ranks (options mi355x_nranks / mi355x_rank or the launcher's
// environment), which device, and the rendez-vous through which rank 0 hands the communicator id to the other ranks.
//
// One Ipopt process per GPU runs the same (deterministic) algorithm; only the KKT factorisation / solves are shared.  The
// reference's only distributed backend initialises its communicator inside the linear-solver adapter as well
// (IpMumpsSolverInterface.cpp:58-75, MPI_Init in the constructor); SPRAL's multi-device knobs are the model for the options
// (IpSpralSolverInterface.cpp:43-67,309-316).
#ifndef IPMI355XCOMMBOOTSTRAP_HPP
#define IPMI355XCOMMBOOTSTRAP_HPP

#include "IpJournalist.hpp"
#include "IpOptionsList.hpp"
#include "mi355x_kkt.h"
#include <string>

namespace Ipopt
{

class Mi355xCommBootstrap
{
public:
   Mi355xCommBootstrap()
      : nranks_opt_(0), rank_opt_(-1), use_shm_(false), ready_(false), generation_(0)
   { }

   /** mi355x_nranks / mi355x_rank / mi355x_comm / mi355x_comm_file / mi355x_subcube (or the launcher's environment: torchrun's RANK /
    *  WORLD_SIZE / LOCAL_RANK, Open MPI's OMPI_COMM_WORLD_*) into kopts.nranks / rank / device / subcube.  A new handle needs a new
    *  communicator: the caller says so with Reset(). */
   void ReadOptions(const OptionsList& options, const std::string& prefix, mi355x_kkt_options& kopts);
   void Reset()
   {
      ready_ = false;
   }
   bool Wanted(const mi355x_kkt_options& kopts) const;
   bool Ready() const
   {
      return ready_;
   }
   /** after mi355x_kkt_analyse: every rank joins the communicator of this set-up (collective over the ranks) */
   bool Setup(mi355x_kkt_handle handle, const mi355x_kkt_options& kopts, const Journalist& jnlst);

private:
   Index nranks_opt_, rank_opt_;
   std::string comm_file_;
   bool use_shm_;                      // mi355x_comm shm: host-staged sum over POSIX shared memory (ranks may share a device)
   bool ready_;
   unsigned int generation_;           // communicators set up through this object so far (part of the rendez-vous record)
};

} // namespace Ipopt
#endif
