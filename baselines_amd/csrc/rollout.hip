// Rollout-side kernels: GAE scan (K2), gather / sf01 (K4/K3), clip+Adam (K8/K9).
#include <string.h>

#include "common.hip.h"

using namespace mrl;

#include <map>
#include <string>
#include <vector>

namespace mrl {
int& gae_lane_form() { static int v = getenv("MRL_GAE_LANE") ? atoi(getenv("MRL_GAE_LANE")) : 1; return v; }
}

extern "C" int mrl_version(void) { return MRL_VERSION; }

// ------------------------------------------------------------------------------------------
// profiler state (the one piece of process-global state in the library; off by default)
// ------------------------------------------------------------------------------------------
namespace {
struct ProfRec { int label; hipEvent_t a, b; double flops, bytes; };
struct ProfLabel { std::string name; long count; double ms, flops, bytes; };
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
std::vector<ProfLabel> g_labels;
std::map<std::string, int> g_label_ids;
int g_open = -1;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void resolve() {
    for (ProfRec& r : g_recs) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        ProfLabel& l = g_labels[r.label];
        l.count += 1; l.ms += ms; l.flops += r.flops; l.bytes += r.bytes;
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
}
}  // namespace

namespace mrl {
bool prof_enabled() { return g_prof_on; }
void prof_begin(const char* label, double flops, double bytes, hipStream_t st) {
    auto it = g_label_ids.find(label);
    int id;
    if (it == g_label_ids.end()) {
        id = (int)g_labels.size();
        g_label_ids[label] = id;
        g_labels.push_back(ProfLabel{label, 0, 0, 0, 0});
    } else id = it->second;
    ProfRec r{id, get_event(), get_event(), flops, bytes};
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    g_open = (int)g_recs.size() - 1;
}
void prof_end(hipStream_t st) {
    if (g_open >= 0) (void)hipEventRecord(g_recs[g_open].b, st);
    g_open = -1;
}
}  // namespace mrl

extern "C" int mrl_prof_enable(int on) {
    if (on) {
        resolve();
        for (ProfLabel& l : g_labels) { l.count = 0; l.ms = l.flops = l.bytes = 0; }
    }
    g_prof_on = on != 0;
    return 0;
}
extern "C" int mrl_prof_num_labels(void) { resolve(); return (int)g_labels.size(); }
extern "C" int mrl_prof_get(int i, char* name, int name_cap, long* count, double* total_ms, double* total_flops,
                            double* total_bytes) {
    resolve();
    if (i < 0 || i >= (int)g_labels.size()) return MRL_EINVAL;
    const ProfLabel& l = g_labels[i];
    if (name && name_cap > 0) { strncpy(name, l.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (count) *count = l.count;
    if (total_ms) *total_ms = l.ms;
    if (total_flops) *total_flops = l.flops;
    if (total_bytes) *total_bytes = l.bytes;
    return 0;
}

extern "C" const char* mrl_strerror(int code) {
    switch (code) {
        case 0: return "ok";
        case MRL_EINVAL: return "mrl: invalid argument";
        case MRL_ENOSPC: return "mrl: workspace too small";
        case MRL_EUNSUP: return "mrl: unsupported configuration";
        case MRL_ECOMM: return "mrl: RCCL communicator error";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "mrl: unknown error";
    }
}

extern "C" int mrl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// GAE.  One workgroup = GAE_E = 16 environments (256 workgroups at num_envs = 4096: one per CU), 256 threads stage
// HBM <-> LDS, the first 16 lanes of wave 0 run the recurrence.  The kernel is LATENCY-bound (nsteps dependent f64
// steps; 8.9 MB of traffic), so the work is arranged for latency: every thread issues ALL its global loads of a time
// chunk before the first LDS write (the old form waited for each element: ~100 dependent round trips to HBM), and
// the recurrence reads 8 steps of operands from LDS ahead of the dependent chain and keeps its outputs in separate
// arrays, so the operand loads never wait behind a store.  Time is processed in chunks of TC steps from the back so
// any nsteps fits; the f64 carry lives in a register.
//
// Reference arithmetic (ppo2/runner.py:58-64 under NumPy promotion rules):
//   gv    = f32(gamma) * V[t+1]                      (f32 product, rounded)
//   delta = ((f64 r[t] + f64(gv) * nnt) - f64 V[t])  (f64)
//   last  = delta + ((gamma*lam) * nnt) * last       (f64; gamma*lam is a host double)
//   adv[t] = f32(last);  ret[t] = adv[t] + V[t]      (f32 add)
// Explicit *_rn intrinsics keep the compiler from contracting any of it into FMAs.
// ------------------------------------------------------------------------------------------
constexpr int GAE_TC = 128;
constexpr int GAE_E = 16;
constexpr int GAE_U = 8;                                        // recurrence steps whose operands are read ahead

__global__ __launch_bounds__(256) void gae_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                  const uint8_t* __restrict__ done,
                                                  const float* __restrict__ last_val,
                                                  const uint8_t* __restrict__ last_done, float gamma_f,
                                                  double gamma_lam, float* __restrict__ adv_out,
                                                  float* __restrict__ ret_out, int T, int N) {
    __shared__ float s_r[GAE_TC][GAE_E];        // rewards
    __shared__ float s_v[GAE_TC + 1][GAE_E];    // values t_lo..t_hi (slot n = V[t_hi])
    __shared__ float s_a[GAE_TC][GAE_E];        // advantages
    __shared__ float s_ret[GAE_TC][GAE_E];
    __shared__ float s_nnt[GAE_TC + 1][GAE_E];  // 1 - done of t_lo+1..t_hi stored at slot (t - t_lo)

    constexpr int IT = ((GAE_TC + 1) * GAE_E + 255) / 256;      // staging elements per thread and chunk
    const int e0 = blockIdx.x * GAE_E;
    const int tid = threadIdx.x;
    const int le = tid % GAE_E, e = e0 + le;
    const bool eok = e < N;
    double carry = 0.0;
    for (int t_hi = T; t_hi > 0; t_hi -= GAE_TC) {
        const int t_lo = max(0, t_hi - GAE_TC);
        const int n = t_hi - t_lo;
        float rr[IT], vv[IT];
        uint8_t dd[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {          // all loads of the chunk in flight before the first LDS write
            const int tt = tid / GAE_E + i * (256 / GAE_E), t = t_lo + tt;
            rr[i] = 0.f; vv[i] = 0.f; dd[i] = 0;
            if (eok && tt <= n) {
                if (tt < n) rr[i] = rew[(long)t * N + e];
                vv[i] = (t < T) ? val[(long)t * N + e] : last_val[e];
                if (tt > 0) dd[i] = (t < T) ? done[(long)t * N + e] : last_done[e];
            }
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int tt = tid / GAE_E + i * (256 / GAE_E);
            if (tt <= n) {
                if (tt < n) s_r[tt][le] = rr[i];
                s_v[tt][le] = vv[i];
                s_nnt[tt][le] = dd[i] ? 0.f : 1.f;
            }
        }
        __syncthreads();
        if (tid < GAE_E && eok) {
            int tt = n - 1;
            for (; tt >= GAE_U - 1; tt -= GAE_U) {
                float r_[GAE_U], v_[GAE_U], v1_[GAE_U], m_[GAE_U];
#pragma unroll
                for (int u = 0; u < GAE_U; ++u) {
                    r_[u] = s_r[tt - u][tid]; v_[u] = s_v[tt - u][tid]; v1_[u] = s_v[tt - u + 1][tid];
                    m_[u] = s_nnt[tt - u + 1][tid];
                }
#pragma unroll
                for (int u = 0; u < GAE_U; ++u) {
                    const double nnt = (double)m_[u];
                    const float gv = __fmul_rn(gamma_f, v1_[u]);
                    const double delta = __dsub_rn(__dadd_rn((double)r_[u], __dmul_rn((double)gv, nnt)), (double)v_[u]);
                    carry = __dadd_rn(delta, __dmul_rn(__dmul_rn(gamma_lam, nnt), carry));
                    const float a = (float)carry;
                    s_a[tt - u][tid] = a;
                    s_ret[tt - u][tid] = __fadd_rn(a, v_[u]);
                }
            }
            for (; tt >= 0; --tt) {
                const double nnt = (double)s_nnt[tt + 1][tid];
                const float v = s_v[tt][tid];
                const float gv = __fmul_rn(gamma_f, s_v[tt + 1][tid]);
                const double delta = __dsub_rn(__dadd_rn((double)s_r[tt][tid], __dmul_rn((double)gv, nnt)), (double)v);
                carry = __dadd_rn(delta, __dmul_rn(__dmul_rn(gamma_lam, nnt), carry));
                const float a = (float)carry;
                s_a[tt][tid] = a;
                s_ret[tt][tid] = __fadd_rn(a, v);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int tt = tid / GAE_E + i * (256 / GAE_E);
            if (eok && tt < n) {
                const long o = (long)(t_lo + tt) * N + e;
                if (adv_out) adv_out[o] = s_a[tt][le];
                ret_out[o] = s_ret[tt][le];
            }
        }
        __syncthreads();
    }
}

// One environment per LANE (round 5): a wave owns 64 consecutive environments, every load / store of a time step is one coalesced row
// segment, the operands of GAE_LU steps are requested before the chain walks them, and nothing goes through LDS.  Same operations
// in the same order per environment as gae_kernel above (bit-identical; the goldens of the reference's Runner.run hold for both);
// only the carry -> carry part of a step (one f64 multiply, one f64 add) is on the dependent chain, delta is formed beside it.
// 19 -> ~7 us at T = 128, N = 4096: all 64 lanes of 64 waves walk chains instead of 16 lanes of 256 workgroups.
constexpr int GAE_LU = 8;
__global__ __launch_bounds__(64) void gae_lane_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                      const uint8_t* __restrict__ done, const float* __restrict__ last_val,
                                                      const uint8_t* __restrict__ last_done, float gamma_f, double gamma_lam,
                                                      float* __restrict__ adv_out, float* __restrict__ ret_out, int T, int N) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= N) return;
    double carry = 0.0;
    float v1 = last_val[e];                     // V[t + 1]
    float m1 = last_done[e] ? 0.f : 1.f;        // 1 - done[t + 1]
    int t = T - 1;
    float rn[GAE_LU], vn[GAE_LU];               // the NEXT block's operands, requested before this block's chain is walked
    uint8_t dn[GAE_LU];
    auto request = [&](int tb) {
#pragma unroll
        for (int u = 0; u < GAE_LU; ++u) {
            const long o = (long)(tb - u) * N + e;
            rn[u] = rew[o]; vn[u] = val[o]; dn[u] = done[o];
        }
    };
    if (t >= GAE_LU - 1) request(t);
    for (; t >= GAE_LU - 1; t -= GAE_LU) {
        float r_[GAE_LU], v_[GAE_LU];
        uint8_t d_[GAE_LU];
#pragma unroll
        for (int u = 0; u < GAE_LU; ++u) { r_[u] = rn[u]; v_[u] = vn[u]; d_[u] = dn[u]; }
        if (t - GAE_LU >= GAE_LU - 1) request(t - GAE_LU);
#pragma unroll
        for (int u = 0; u < GAE_LU; ++u) {
            const double nnt = (double)m1;
            const float gv = __fmul_rn(gamma_f, v1);
            const double delta = __dsub_rn(__dadd_rn((double)r_[u], __dmul_rn((double)gv, nnt)), (double)v_[u]);
            carry = __dadd_rn(delta, __dmul_rn(__dmul_rn(gamma_lam, nnt), carry));
            const float a = (float)carry;
            const long o = (long)(t - u) * N + e;
            if (adv_out) adv_out[o] = a;
            ret_out[o] = __fadd_rn(a, v_[u]);
            v1 = v_[u];
            m1 = d_[u] ? 0.f : 1.f;             // done[t] gates step t - 1
        }
    }
    for (; t >= 0; --t) {
        const long o = (long)t * N + e;
        const float r = rew[o], v = val[o];
        const uint8_t d = done[o];
        const double nnt = (double)m1;
        const float gv = __fmul_rn(gamma_f, v1);
        const double delta = __dsub_rn(__dadd_rn((double)r, __dmul_rn((double)gv, nnt)), (double)v);
        carry = __dadd_rn(delta, __dmul_rn(__dmul_rn(gamma_lam, nnt), carry));
        const float a = (float)carry;
        if (adv_out) adv_out[o] = a;
        ret_out[o] = __fadd_rn(a, v);
        v1 = v;
        m1 = d ? 0.f : 1.f;
    }
}

extern "C" int mrl_gae(const float* rew, const float* val, const uint8_t* done, const float* last_val,
                       const uint8_t* last_done, double gamma, double lam, float* adv_out, float* ret_out,
                       int T, int N, void* stream) {
    if (T <= 0 || N <= 0 || !rew || !val || !done || !last_val || !last_done || !ret_out) return MRL_EINVAL;
    ProfScope ps("gae", 0.0, 17.0 * T * N + 9.0 * N, (hipStream_t)stream);
    // option "gae_lane" [MRL_GAE_LANE, 1]; 0: the LDS-staged kernel at every size (A/B; bit-identical)
    // a lane per environment needs a wave's worth of environments; smaller vector envs keep the 16-per-workgroup form
    if (mrl::gae_lane_form() && N >= 64) {
        hipLaunchKernelGGL(gae_lane_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, rew, val, done, last_val, last_done,
                           (float)gamma, gamma * lam, adv_out, ret_out, T, N);
    } else {
        dim3 grid((N + GAE_E - 1) / GAE_E);
        hipLaunchKernelGGL(gae_kernel, grid, dim3(256), 0, (hipStream_t)stream, rew, val, done, last_val, last_done,
                           (float)gamma, gamma * lam, adv_out, ret_out, T, N);
    }
    MRL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Row gather / sf01.  Rows are moved 16 B per lane when row_bytes % 16 == 0, else 4 B / 1 B.
// ------------------------------------------------------------------------------------------
template <typename V>
__global__ __launch_bounds__(256) void gather_rows_kernel(const V* __restrict__ src, const int64_t* __restrict__ idx,
                                                          V* __restrict__ dst, long B, int T, int N, int rowv,
                                                          int mode /*0 gather idx, 1 sf01*/) {
    long total = B * (long)rowv;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += (long)gridDim.x * 256L) {
        long b = q / rowv;
        int c = (int)(q - b * rowv);
        long i = (mode == 0) ? idx[b] : b;
        long srow = envmajor_to_row(i, T, N);
        dst[q] = src[srow * rowv + c];
    }
}

static int launch_gather(const void* src, const int64_t* idx, void* dst, long B, int T, int N, int row_bytes,
                         int mode, hipStream_t st) {
    if (B <= 0) return 0;
    if (row_bytes <= 0 || T <= 0 || N <= 0) return MRL_EINVAL;
    bool a16 = (row_bytes % 16 == 0) && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    bool a4 = (row_bytes % 4 == 0) && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0);
    int unit = a16 ? 16 : (a4 ? 4 : 1);
    int rowv = row_bytes / unit;
    long total = B * (long)rowv;
    int blocks = (int)min((total + 255) / 256, (long)8192);
    if (unit == 16)
        hipLaunchKernelGGL(gather_rows_kernel<uint4>, dim3(blocks), dim3(256), 0, st, (const uint4*)src, idx,
                           (uint4*)dst, B, T, N, rowv, mode);
    else if (unit == 4)
        hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(blocks), dim3(256), 0, st, (const uint32_t*)src, idx,
                           (uint32_t*)dst, B, T, N, rowv, mode);
    else
        hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (const uint8_t*)src, idx,
                           (uint8_t*)dst, B, T, N, rowv, mode);
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrl_gather_rows(const void* src, const int64_t* idx, void* dst, int B, int T, int N, int row_bytes,
                               void* stream) {
    if (!src || !idx || !dst) return MRL_EINVAL;
    return launch_gather(src, idx, dst, B, T, N, row_bytes, 0, (hipStream_t)stream);
}

extern "C" int mrl_sf01(const void* src, void* dst, int T, int N, int row_bytes, void* stream) {
    if (!src || !dst) return MRL_EINVAL;
    return launch_gather(src, nullptr, dst, (long)T * N, T, N, row_bytes, 1, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// clip_by_global_norm + Adam (TF-1 ApplyAdam form), 16-byte accesses:
//   [1) sumsq partials (f64, fixed order) of g/total_weight -- skipped when the producer of the gradient already
//       left them behind: reduce_slabs in model.hip emits the partial sums of squares of what it writes, so the
//       gradient is not read a second time for its norm (mrl_model_train_step);]
//   2) every block re-reduces the partials in the same order -> identical norm everywhere, then updates its
//      slice:  m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= m*alpha/(sqrt(v)+eps).
// Algorithmic traffic 28 B/param (read p,g,m,v; write p,m,v) [+ 4 B/param for a separate norm pass].
// ------------------------------------------------------------------------------------------
constexpr int ADAM_MAX_PART = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long P, float total_weight,
                                                    double* __restrict__ part) {
    __shared__ double sh[4];
    double s = 0.0;
    const long P4 = P >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < P4; i += (long)gridDim.x * 256L) {
        float4 x = g4[i];
        if (total_weight != 1.f) { x.x /= total_weight; x.y /= total_weight; x.z /= total_weight; x.w /= total_weight; }
        s += ((double)x.x * (double)x.x + (double)x.y * (double)x.y) + ((double)x.z * (double)x.z + (double)x.w * (double)x.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(P & 3)) {          // tail
        float x = g[P4 * 4 + threadIdx.x];
        if (total_weight != 1.f) x = x / total_weight;
        s += (double)x * (double)x;
    }
    double r = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__device__ __forceinline__ void adam_elem(float& p, float& g, float& m, float& v, float alpha, float omb1, float omb2,
                                          float eps, float total_weight, float scale) {
    float x = g;
    if (total_weight != 1.f) x = x / total_weight;
    x = x * scale;
    g = x;          // the clipped, averaged gradient (model.py:112 self.grads)
    m = m + (x - m) * omb1;
    v = v + (x * x - v) * omb2;
    p = p - (m * alpha) / (sqrtf(v) + eps);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long P, float alpha, float beta1,
                                                   float beta2, float eps, float max_grad_norm, float total_weight,
                                                   const double* __restrict__ part, int npart,
                                                   float* __restrict__ gnorm_out, const float* __restrict__ alpha_dev) {
    __shared__ double sh[4];
    __shared__ float s_scale;
    float scale = 1.f;
    if (alpha_dev) alpha = alpha_dev[0];      // step size kept in device memory (replayable launch graphs)
    if (max_grad_norm >= 0.f) {
        double s = 0.0;
        for (int i = threadIdx.x; i < npart; i += 256) s += part[i];
        double tot = block_sum_256(s, sh);
        if (threadIdx.x == 0) {
            float gn = (float)sqrt(tot);
            // tf.clip_by_global_norm: scale = clip_norm * min(1/global_norm, 1/clip_norm)
            s_scale = max_grad_norm * fminf(1.f / gn, 1.f / max_grad_norm);
            if (gnorm_out && blockIdx.x == 0) gnorm_out[0] = gn;
        }
        __syncthreads();
        scale = s_scale;
    }
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    const long P4 = P >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < P4; i += (long)gridDim.x * 256L) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam_elem(pp.x, gg.x, mm.x, vv.x, alpha, omb1, omb2, eps, total_weight, scale);
        adam_elem(pp.y, gg.y, mm.y, vv.y, alpha, omb1, omb2, eps, total_weight, scale);
        adam_elem(pp.z, gg.z, mm.z, vv.z, alpha, omb1, omb2, eps, total_weight, scale);
        adam_elem(pp.w, gg.w, mm.w, vv.w, alpha, omb1, omb2, eps, total_weight, scale);
        g4[i] = gg; m4[i] = mm; v4[i] = vv; p4[i] = pp;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(P & 3)) {          // tail
        const long i = P4 * 4 + threadIdx.x;
        float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        adam_elem(pp, gg, mm, vv, alpha, omb1, omb2, eps, total_weight, scale);
        g[i] = gg; m[i] = mm; v[i] = vv; p[i] = pp;
    }
}

// MicrobatchedModel (microbatched_model.py:57-66): every microbatch gradient is averaged over ranks and clipped by
// ITS OWN global norm (the reference sums `self.grads`, which model.py:105-112 defines post-clip), then summed.
__global__ __launch_bounds__(256) void clip_accumulate_kernel(const float* __restrict__ g, float* __restrict__ acc, long P,
                                                              float max_grad_norm, float total_weight, int first,
                                                              const double* __restrict__ part, int npart) {
    __shared__ double sh[4];
    __shared__ float s_scale;
    float scale = 1.f;
    if (max_grad_norm >= 0.f) {
        double s = 0.0;
        for (int i = threadIdx.x; i < npart; i += 256) s += part[i];
        double tot = block_sum_256(s, sh);
        if (threadIdx.x == 0) {
            float gn = (float)sqrt(tot);
            s_scale = max_grad_norm * fminf(1.f / gn, 1.f / max_grad_norm);
        }
        __syncthreads();
        scale = s_scale;
    }
    for (long i = blockIdx.x * 256L + threadIdx.x; i < P; i += (long)gridDim.x * 256L) {
        float x = g[i];
        if (total_weight != 1.f) x = x / total_weight;
        x = x * scale;
        acc[i] = first ? x : acc[i] + x;
    }
}

static inline bool vec16(const void* p) { return (uintptr_t)p % 16 == 0; }

extern "C" int mrl_clip_accumulate(const float* grads, float* acc, long P, float max_grad_norm, float total_weight,
                                   int first, void* scratch, void* stream) {
    if (!grads || !acc || P <= 0 || !scratch || total_weight <= 0.f) return MRL_EINVAL;
    if (!vec16(grads)) return MRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int blocks = (int)min((P + 1023) / 1024, (long)ADAM_MAX_PART);
    ProfScope ps("clip+accumulate", 0.0, (max_grad_norm >= 0.f ? 16.0 : 12.0) * P, st);
    if (max_grad_norm >= 0.f) {
        hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, st, grads, P, total_weight, (double*)scratch);
        MRL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(clip_accumulate_kernel, dim3(blocks), dim3(256), 0, st, grads, acc, P, max_grad_norm, total_weight,
                       first, (const double*)scratch, blocks);
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t mrl_adam_scratch_bytes(long P) {
    (void)P;
    return ADAM_MAX_PART * sizeof(double);
}

namespace mrl {
// `ready_part` / `ready_npart`: sum-of-squares partials of the (single-rank, un-averaged) gradient left behind by its
// producer; nullptr: computed here with one extra pass over the gradient into `scratch`.
int adam_clip_apply(float* params, float* grads, float* m, float* v, long P, float alpha, const float* alpha_dev,
                    float beta1, float beta2, float eps, float max_grad_norm, float total_weight, float* gnorm_out,
                    void* scratch, const double* ready_part, int ready_npart, hipStream_t st) {
    if (!params || !grads || !m || !v || P <= 0 || total_weight <= 0.f) return MRL_EINVAL;
    if (!vec16(params) || !vec16(grads) || !vec16(m) || !vec16(v)) return MRL_EINVAL;
    if (max_grad_norm >= 0.f && !ready_part && !scratch) return MRL_EINVAL;
    const bool fused = ready_part != nullptr && max_grad_norm >= 0.f;
    ProfScope ps("clip+adam", 0.0, (max_grad_norm >= 0.f && !fused ? 32.0 : 28.0) * P, st);
    const double* part = ready_part;
    int npart = ready_npart;
    if (max_grad_norm >= 0.f && !ready_part) {
        npart = (int)min((P + 4095) / 4096, (long)ADAM_MAX_PART);
        hipLaunchKernelGGL(sumsq_kernel, dim3(npart), dim3(256), 0, st, grads, P, total_weight, (double*)scratch);
        MRL_LAUNCH_CHECK();
        part = (const double*)scratch;
    }
    const int blocks = (int)min((P / 4 + 255) / 256 + 1, (long)2048);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, params, grads, m, v, P, alpha, beta1, beta2, eps,
                       max_grad_norm, total_weight, part, npart, gnorm_out, alpha_dev);
    MRL_LAUNCH_CHECK();
    return 0;
}
}  // namespace mrl

extern "C" int mrl_adam_clip_step(float* params, float* grads, float* m, float* v, long P, float alpha, float beta1,
                                  float beta2, float eps, float max_grad_norm, float total_weight, float* gnorm_out,
                                  void* scratch, void* stream) {
    return adam_clip_apply(params, grads, m, v, P, alpha, nullptr, beta1, beta2, eps, max_grad_norm, total_weight,
                           gnorm_out, scratch, nullptr, 0, (hipStream_t)stream);
}

extern "C" int mrl_adam_clip_step_dev(float* params, float* grads, float* m, float* v, long P, const float* alpha_dev,
                                      float beta1, float beta2, float eps, float max_grad_norm, float total_weight,
                                      float* gnorm_out, void* scratch, void* stream) {
    if (!alpha_dev) return MRL_EINVAL;
    return adam_clip_apply(params, grads, m, v, P, 0.f, alpha_dev, beta1, beta2, eps, max_grad_norm, total_weight,
                           gnorm_out, scratch, nullptr, 0, (hipStream_t)stream);
}
