// shared host/device helpers for libmrl
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mrl.h"

#define MRL_HIP_CHECK(expr)                           \
    do {                                              \
        hipError_t _e = (expr);                       \
        if (_e != hipSuccess) return (int)_e;         \
    } while (0)

#define MRL_LAUNCH_CHECK()                            \
    do {                                              \
        hipError_t _e = hipGetLastError();            \
        if (_e != hipSuccess) return (int)_e;         \
    } while (0)

namespace mrl {

// deterministic block-wide sum of doubles (256 threads), result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* sh /* >= 4 doubles */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    return r;
}

// env-major flat index (runner.py:69-74) -> time-major storage row
__device__ __forceinline__ long envmajor_to_row(long i, int T, int N) { return (i % T) * (long)N + i / T; }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Kernels that need more than 64 KB of dynamic LDS raise their limit once PER DEVICE (the attribute belongs to the device's copy of the
// function): one driver call per (device, kernel), none on the hot path or during graph capture afterwards; safe from several host
// threads.  (Through round 5 every launcher kept a process-wide `static bool`: a second HIP device would have skipped the call.)
hipError_t raise_lds_limit(const void* kern, int bytes = 160 * 1024);
// option "gae_lane" (rollout.hip owns it; mrl_set_option in model.hip writes it)
int& gae_lane_form();

// ---- optional HIP-event profiler (mrl_prof_*): per-label launch count / time / algorithmic work.
// Events are recorded on the launch stream itself; nothing synchronises until the report is read.
bool prof_enabled();
void prof_begin(const char* label, double flops, double bytes, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
    hipStream_t st; bool on;
    ProfScope(const char* label, double flops, double bytes, hipStream_t s) : st(s), on(prof_enabled()) {
        if (on) prof_begin(label, flops, bytes, st);
    }
    ~ProfScope() { if (on) prof_end(st); }
};


// clip_by_global_norm + TF-1 Adam (rollout.hip); ready_part: sum-of-squares partials left behind by the gradient's producer
int adam_clip_apply(float* params, float* grads, float* m, float* v, long P, float alpha, const float* alpha_dev,
                    float beta1, float beta2, float eps, float max_grad_norm, float total_weight, float* gnorm_out,
                    void* scratch, const double* ready_part, int ready_npart, hipStream_t st);

}  // namespace mrl
