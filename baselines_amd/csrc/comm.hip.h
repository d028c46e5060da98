// internal interface of comm.hip (RCCL communicator) used by the learner orchestration in model.hip
#pragma once
#include <algorithm>

#include "common.hip.h"

namespace mrl {
int comm_allreduce_async(mrl_comm* c, float* g, long n, float weight, hipStream_t compute);
int comm_join(mrl_comm* c, hipStream_t compute);
}  // namespace mrl
