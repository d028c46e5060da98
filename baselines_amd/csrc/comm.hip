// Data-parallel collectives inside the boundary: RCCL over xGMI, one communicator per process (= per GPU).
//
// Replaces the reference's host-staged MPI exchange (paths relative to baselines/):
//   common/mpi_adam_optimizer.py:21,39-40   flat grad * rank weight -> Allreduce(SUM) -> / sum of weights
//   common/mpi_util.py:15-26                sync_from_root: Bcast of every global variable from rank 0
//   common/mpi_adam_optimizer.py:53-68      check_synced
// The reference copies the gradient D->H, reduces over MPI and copies it back for every minibatch step; here the
// all-reduce runs on a communication stream of the GPU, ordered against the compute stream by events, so that the
// gradient of the large fc1 layer (95 % of the parameters, ready first in the backward pass) travels while the
// convolution backward kernels still run (model.hip: model_grad_range).
//
// librccl is bound lazily with dlopen (first the copy already mapped into the process, e.g. the one PyTorch-ROCm
// ships -- two different RCCL builds in one process would each bring their own HIP runtime bindings): a host that
// never creates a communicator never loads it, and libmrl.so has no link-time dependency on it.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <rccl/rccl.h>

#include "comm.hip.h"

using namespace mrl;

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
char g_rccl_err[256] = "";

bool rccl_load() {
    Rccl& r = g_rccl;
    if (r.ok) return true;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
        if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // already in the process (PyTorch-ROCm's copy)
    for (const char* n : names)
        if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!r.h) {
        snprintf(g_rccl_err, sizeof g_rccl_err, "librccl not found: %s", dlerror());
        return false;
    }
#define MRL_SYM(field, name)                                                              \
    *(void**)(&r.field) = dlsym(r.h, name);                                               \
    if (!r.field) { snprintf(g_rccl_err, sizeof g_rccl_err, "librccl lacks %s", name); return false; }
    MRL_SYM(GetUniqueId, "ncclGetUniqueId")
    MRL_SYM(CommInitRank, "ncclCommInitRank")
    MRL_SYM(CommDestroy, "ncclCommDestroy")
    MRL_SYM(AllReduce, "ncclAllReduce")
    MRL_SYM(Broadcast, "ncclBroadcast")
    MRL_SYM(GetErrorString, "ncclGetErrorString")
#undef MRL_SYM
    r.ok = true;
    return true;
}

int nccl_rc(ncclResult_t e) {
    if (e == ncclSuccess) return 0;
    snprintf(g_rccl_err, sizeof g_rccl_err, "rccl: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error");
    return MRL_ECOMM;
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, float s) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) x[i] *= s;
}
}  // namespace

struct mrl_comm {
    ncclComm_t comm;
    int nranks, rank;
    hipStream_t side;                 // communication stream (non-blocking)
    hipEvent_t ev_ready, ev_done;     // compute -> side, side -> compute
    bool side_busy;
};

extern "C" const char* mrl_comm_last_error(void) { return g_rccl_err; }

extern "C" int mrl_comm_unique_id(void* id_out) {
    if (!id_out) return MRL_EINVAL;
    if (!rccl_load()) return MRL_ECOMM;
    static_assert(sizeof(ncclUniqueId) == MRL_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    int rc = nccl_rc(g_rccl.GetUniqueId(&id));
    if (rc) return rc;
    memcpy(id_out, &id, sizeof id);
    return 0;
}

extern "C" int mrl_comm_create(const void* id, int nranks, int rank, mrl_comm** out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return MRL_EINVAL;
    if (!rccl_load()) return MRL_ECOMM;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    mrl_comm* c = new mrl_comm();
    c->nranks = nranks; c->rank = rank; c->side_busy = false;
    int rc = nccl_rc(g_rccl.CommInitRank(&c->comm, nranks, uid, rank));     // on the caller's current device
    if (rc) { delete c; return rc; }
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) { g_rccl.CommDestroy(c->comm); delete c; return (int)e; }
    *out = c;
    return 0;
}

extern "C" void mrl_comm_destroy(mrl_comm* c) {
    if (!c) return;
    (void)hipStreamSynchronize(c->side);
    if (g_rccl.ok) g_rccl.CommDestroy(c->comm);
    (void)hipEventDestroy(c->ev_ready);
    (void)hipEventDestroy(c->ev_done);
    (void)hipStreamDestroy(c->side);
    delete c;
}

extern "C" int mrl_comm_size(const mrl_comm* c) { return c ? c->nranks : 0; }
extern "C" int mrl_comm_rank(const mrl_comm* c) { return c ? c->rank : -1; }

extern "C" int mrl_allreduce_grads(mrl_comm* c, float* grads, long P, void* stream) {
    if (!c || !grads || P <= 0) return MRL_EINVAL;
    ProfScope ps("allreduce", 0.0, 4.0 * P, (hipStream_t)stream);
    return nccl_rc(g_rccl.AllReduce(grads, grads, (size_t)P, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
}

extern "C" int mrl_broadcast_state(mrl_comm* c, void* buf, size_t nbytes, int root, void* stream) {
    if (!c || !buf || root < 0 || root >= c->nranks) return MRL_EINVAL;
    if (nbytes == 0) return 0;
    return nccl_rc(g_rccl.Broadcast(buf, buf, nbytes, ncclUint8, root, c->comm, (hipStream_t)stream));
}

namespace mrl {
// Issue an all-reduce(sum) of g[0..n) on the communication stream as soon as everything launched on `compute` so far
// has finished; `weight` != 1 scales the slice first (mpi_adam_optimizer.py:21 `flat_grad * mpi_rank_weight`).
int comm_allreduce_async(mrl_comm* c, float* g, long n, float weight, hipStream_t compute) {
    if (n <= 0) return 0;
    MRL_HIP_CHECK(hipEventRecord(c->ev_ready, compute));
    MRL_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_ready, 0));
    if (weight != 1.f) {
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, c->side, g, n, weight);
        MRL_LAUNCH_CHECK();
    }
    int rc = nccl_rc(g_rccl.AllReduce(g, g, (size_t)n, ncclFloat32, ncclSum, c->comm, c->side));
    c->side_busy = true;
    return rc;
}
// Everything launched on `compute` after this call sees the reduced gradient.
int comm_join(mrl_comm* c, hipStream_t compute) {
    if (!c->side_busy) return 0;
    MRL_HIP_CHECK(hipEventRecord(c->ev_done, c->side));
    MRL_HIP_CHECK(hipStreamWaitEvent(compute, c->ev_done, 0));
    c->side_busy = false;
    return 0;
}
}  // namespace mrl
