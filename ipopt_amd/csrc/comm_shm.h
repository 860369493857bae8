// comm_shm.h -- a host-staged communicator over POSIX shared memory (one node; ranks may SHARE a device).
// The third communicator kind behind mi355x_kkt_set_comm_* (include/mi355x_kkt.h): RCCL refuses two ranks on one GPU, so
// bring-up, the one-GPU test box and CPU-launcher smoke runs get the one collective the solver needs -- an in-place sum over
// a range of ranks -- through a shared segment: every rank copies its piece to its slot, the ranks sum the slots in RANK ORDER
// (so all of them hold bitwise the same result) and copy it back.  Not a fast path: production multi-GPU is RCCL over xGMI.
#pragma once
#include <cstdint>
#include <string>

namespace mi355x {

struct ShmComm;   // opaque

// rank 0: creates the segment and writes its (unique) name, zero-terminated, into out128
bool shm_comm_create(int nranks, void* out128, std::string& err);
// every rank (rank 0 too): attaches, waits until all nranks have attached (rank 0 then unlinks the name: nothing is left behind
// whatever happens later).  Returns nullptr + err on failure / time-out.
ShmComm* shm_comm_attach(const void* id128, int rank, int nranks, std::string& err);
void shm_comm_destroy(ShmComm* c);
// rank 0, when the id could not be handed out after all (the rendez-vous file could not be written): unlink and unmap a segment nobody attached to
void shm_comm_discard(const void* id128);
// the two callbacks of include/mi355x_kkt.h (ctx = ShmComm*)
int shm_comm_allreduce(void* ctx, void* dptr, int64_t count, int dtype, void* hip_stream);
int shm_comm_allreduce_range(void* ctx, void* dptr, int64_t count, int dtype, void* hip_stream, int rank_lo, int nranks_in_range);

}  // namespace mi355x
