// RCCL inside the library (round 5): the data-parallel all-reduces of the two gradient buckets are enqueued by the ENGINE on its own
// streams -- inside the captured step graph -- instead of by torch.distributed between graph launches (reference counterpart: the
// gradient reduce of nn.DataParallel, Solver.py:33-35).  librccl.so.1 is dlopen'ed on first use (the process-wide copy torch has already
// loaded, when there is one: same soname), so the library neither links RCCL nor loads it in single-GPU runs.
#pragma once
#include <cstddef>

#include "common.h"

namespace mimrl {

int comm_unique_id(void* out128);                                         // rank 0: ncclGetUniqueId (128 bytes)
int comm_init(void** comm, const void* id128, int world, int rank);       // ncclCommInitRank on the current device (collective over the ranks)
int comm_allreduce_sum(void* comm, float* buf, size_t n, hipStream_t s);  // in place, fp32 sum, stream-ordered (capturable)
int comm_allreduce_sum_bf16(void* comm, void* buf, size_t n, hipStream_t s);   // the same on n bf16 values (RCCL sums them in bf16)
int comm_destroy(void* comm);

}  // namespace mimrl
