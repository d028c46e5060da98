// Fused PPO2 minibatch gradient for the MLP policy/value networks (common/models.py:74-103 `mlp`,
// 2 x 64 tanh; common/policies.py:43-64 heads; ppo2/model.py:57-91 loss) on gfx950.
//
// Why a fused kernel: the MuJoCo-shaped config runs 320 optimizer steps per update on 4096-sample
// minibatches -- ~1 GFLOP each.  As ~22 separate launches (layer GEMMs, heads, slab reductions) a step
// takes ~470 us of latency chains (measured, profiles/); the arithmetic is ~10 us.  Here ONE workgroup owns
// a 32-sample tile and carries it through the whole step without leaving the CU:
//   P0 gather the observation rows into LDS            P4 head gradients, dz of the last hidden layer (VALU)
//   P1 fc0 forward, both nets (one 32x32 MFMA tile/wave) P5 fc1 weight gradient + data gradient (MFMA)
//   P2 fc1 forward (MFMA)                               P6 fc0 weight gradient (MFMA, 12 tiles per wave)
//   P3 heads + PPO loss + closed-form gradients (VALU)
// Activations never touch HBM; weights stream from L2 straight into MFMA B fragments; every workgroup
// writes one partial-gradient slab in the flat parameter layout, combined in fixed order by reduce_slabs
// (deterministic).  Arithmetic order of the head / loss code is the same as heads_train_kernel's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hip.h"

namespace mrl {

struct MlpStepArgs {
    // parameter layout (flat offsets; index 0 = policy net, 1 = value net when nets == 2)
    long w0[2], b0[2], w1[2], b1[2];
    long wpi, bpi, logstd, wvf, bvf;      // logstd < 0: categorical
    int K0, nact, nets, pd_kind;
    long P;
    // data
    const float* params; const float* obs; const int32_t* srow;     // srow[b]: storage row (nullptr: row0 + b)
    const void* actions; const float* returns; const float* values; const float* neglogp;
    const float* advstat; float cliprange, ent_coef, vf_coef, invB;
    int B;
    // advstat == nullptr: every workgroup computes the minibatch advantage statistics itself (model.py:136-139) over
    // the Bstat samples stat_idx[0..Bstat) (env-major indices; nullptr: rows 0..Bstat-1) of stat_ret / stat_val, in the
    // same fixed order everywhere -- identical results in all workgroups, two launches fewer per step.
    // tile_idx != nullptr (with advstat == nullptr): env-major indices of THIS call's samples, translated here.
    const float* stat_ret; const float* stat_val; const int64_t* stat_idx; const int64_t* tile_idx; int Bstat, T, N;
    float* part;           // [ntiles][P]
    double* spart;         // [ntiles][5]
    long long* dbg;        // optional [8] phase timestamps of workgroup 0 (s_memtime), nullptr normally
};

constexpr int MLP_NH = 64;             // hidden width
constexpr int MLP_LD = MLP_NH + 4;     // LDS row stride of the 64-wide activations
// Workgroup = 8 waves (round 3; 4 before).  The step is a chain of LATENCY-bound phases on a 32-sample tile -- 128 workgroups
// for a 4096-sample minibatch, half the chip idle whatever the workgroup size -- so the waves of a workgroup are what
// shortens a phase: the k range of the two forward GEMMs is split over two wave groups (partial accumulators combined
// through LDS in a fixed order), the 12 + 48 weight-gradient tiles and every thread-strided loop are spread over twice
// the waves.

// deterministic block-wide sum of doubles (MLP_NW waves), result valid in thread 0
template <int MLP_NW>
__device__ __forceinline__ double mlp_block_sum(double v, double* sh /* >= MLP_NW doubles */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < MLP_NW; ++q) r += sh[q];
    }
    return r;
}

#define MRL_MLP_HALF_LOG_2PI 0.9189385332046727f
#define MRL_MLP_HALF_LOG_2PIE 1.4189385332046727f

template <int MLP_NT>
__global__ __launch_bounds__(MLP_NT) void mlp_step_kernel(MlpStepArgs a) {
    constexpr int MLP_NW = MLP_NT / 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int K0 = a.K0, KB0 = (K0 + 7) / 8, KP = KB0 * 8 + 4;
    const int nets = a.nets, nact = a.nact;
    const bool shared = nets == 1;
    const int s0 = blockIdx.x * 32;
    const float* P = a.params;
    float* slab = a.part + (long)blockIdx.x * a.P;
    // NOTE: never index a.w0[] etc. with a runtime value -- a dynamically indexed kernel-argument array is
    // spilled to memory and re-loaded (global_load + s_waitcnt vmcnt(0)) in front of every use.
    auto W0 = [&](int n) { return n ? a.w0[1] : a.w0[0]; };
    auto B0 = [&](int n) { return n ? a.b0[1] : a.b0[0]; };
    auto W1 = [&](int n) { return n ? a.w1[1] : a.w1[0]; };
    auto B1 = [&](int n) { return n ? a.b1[1] : a.b1[0]; };
    auto stamp = [&](int k) { if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[k] = (long long)__builtin_readcyclecounter(); };
    stamp(0);

    // ---- LDS carve
    float* obs_s = sm;                                   // [32][KP]
    float* h0_s = obs_s + 32 * KP;                       // [nets][32][MLP_LD]
    float* h1_s = h0_s + nets * 32 * MLP_LD;
    float* dz1_s = h1_s + nets * 32 * MLP_LD;
    float* dz0_s = dz1_s + nets * 32 * MLP_LD;
    float* pi_s = dz0_s + nets * 32 * MLP_LD;            // [32][32] pdparam (mean / logits)
    float* dpi_s = pi_s + 32 * 32;                       // [32][32]
    float* dls_s = dpi_s + 32 * 32;                      // [32][32]
    float* v_s = dls_s + 32 * 32;                        // [32]
    float* dv_s = v_s + 32;                              // [32]
    float* wpi_s = dv_s + 32;                            // [64][nact] policy head weights
    float* wvf_s = wpi_s + 64 * 32;                      // [64] value head weights
    long* row_s = reinterpret_cast<long*>(wvf_s + 64);   // [32] storage rows (-1: beyond the minibatch)
    float* kp_s = reinterpret_cast<float*>(row_s + 32);  // [4 waves][16][64] partial accumulators of the second k half

    // ---- P0: rows + observation tile
    if (tid < 32) {
        const int b = s0 + tid;
        long r = -1;
        if (b < a.B) r = a.tile_idx ? envmajor_to_row(a.tile_idx[b], a.T, a.N) : (a.srow ? (long)a.srow[b] : (long)b);
        row_s[tid] = r;
    }
    // ---- P0': advantage statistics of the whole minibatch (f64 accumulation, fixed order)
    float adv_mean, adv_sd;
    if (!a.advstat) {
        double* red = reinterpret_cast<double*>(pi_s);       // LDS scratch; pi_s is first written in P3a
        double s1 = 0.0, s2 = 0.0;
        // batches of 8 samples per thread: all index loads, then all gathers in flight together (one wave per SIMD here:
        // every dependent load would otherwise expose its full latency, 2 per sample)
        for (int b0 = tid; b0 < a.Bstat; b0 += 8 * MLP_NT) {
            long r[8];
            float rv[8], vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = min(b0 + u * MLP_NT, a.Bstat - 1);
                r[u] = a.stat_idx ? (long)a.stat_idx[b] : (long)b;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (a.stat_idx) r[u] = envmajor_to_row(r[u], a.T, a.N);
                rv[u] = a.stat_ret[r[u]];
                vv[u] = a.stat_val[r[u]];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b0 + u * MLP_NT < a.Bstat) {
                    const float x = __fsub_rn(rv[u], vv[u]);
                    s1 += (double)x;
                    s2 += (double)x * (double)x;
                }
        }
        const double t1 = mlp_block_sum<MLP_NW>(s1, red);
        const double t2 = mlp_block_sum<MLP_NW>(s2, red + MLP_NW);
        if (tid == 0) {
            const double mean = t1 / a.Bstat;
            double var = t2 / a.Bstat - mean * mean;
            if (var < 0) var = 0;
            red[2 * MLP_NW] = (double)(float)mean;
            red[2 * MLP_NW + 1] = (double)(float)sqrt(var);
        }
        __syncthreads();
        adv_mean = (float)red[2 * MLP_NW];
        adv_sd = (float)red[2 * MLP_NW + 1] + 1e-8f;
    } else {
        adv_mean = a.advstat[0];
        adv_sd = a.advstat[1] + 1e-8f;
    }
    __syncthreads();
    // one wave per SIMD: every memory latency is exposed, so loads are issued in batches of 8 before use
    {
        const int nv = 32 * (KP / 4);
        for (int e0 = tid; e0 < nv; e0 += 8 * MLP_NT) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * MLP_NT;
                v[u] = f4zero();
                if (e < nv) {
                    const int s = e / (KP / 4), c4 = (e % (KP / 4)) * 4;
                    const long r = row_s[s];
                    if (r >= 0 && c4 + 3 < K0) v[u] = *reinterpret_cast<const float4*>(a.obs + r * K0 + c4);
                    else if (r >= 0 && c4 < K0) v[u] = load_partial(a.obs + r * K0 + c4, K0 - c4);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * MLP_NT;
                if (e < nv) {
                    const int s = e / (KP / 4), c4 = (e % (KP / 4)) * 4;
                    *reinterpret_cast<float4*>(obs_s + s * KP + c4) = v[u];
                }
            }
        }
        // small weights -> LDS (coalesced): pi head, value head, biases, logstd
        for (int e = tid; e < 64 * nact; e += MLP_NT) wpi_s[e] = P[a.wpi + e];
        for (int e = tid; e < 64; e += MLP_NT) wvf_s[e] = P[a.wvf + e];
    }
    __syncthreads();

    stamp(1);
    // ---- P1: fc0 forward.  wave -> (k half, net, 32-column tile); nets == 1: waves 2, 3, 6, 7 idle
    {
        const int wq = wave & 3, khalf = wave >> 2;
        const int net = wq >> 1, n0 = (wq & 1) * 32;
        if (net < nets) {
            const float* W = P + W0(net);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* arow = obs_s + i * KP + 4 * h;
            const float* wcol = W + n0 + i;
            // B fragments come straight from L2: issue the 16 loads of the NEXT group of 4 k-blocks before the
            // 16 MFMAs of the current one (unconditional loads with a clamped row, zeroed by select)
            auto loadg = [&](int g, float (&dst)[16]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k = 8 * (4 * g + u) + 4 * h + q;
                        const float w = wcol[(long)min(k, K0 - 1) * MLP_NH];
                        dst[4 * u + q] = k < K0 ? w : 0.f;
                    }
            };
            auto mmag = [&](int g, const float (&fbv)[16]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kb = 4 * g + u;
                    if (kb < KB0) {
                        const float4 fa = *reinterpret_cast<const float4*>(arow + 8 * kb);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fbv[4 * u + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fbv[4 * u + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fbv[4 * u + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fbv[4 * u + 3], acc, 0, 0, 0);
                    }
                }
            };
            // groups of 4 k blocks: [0, NGH) for the first wave group, [NGH, NGT) for the second
            const int NGT = (KB0 + 3) / 4, NGH = MLP_NW > 4 ? (NGT + 1) / 2 : NGT;
            const int gb = khalf ? NGH : 0, NG = khalf ? NGT : NGH;
            float fb0[16], fb1[16];
            if (gb < NG) loadg(gb, fb0);
            for (int g = gb; g < NG; g += 2) {
                if (g + 1 < NG) loadg(g + 1, fb1);
                mmag(g, fb0);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < NG) loadg(g + 2, fb0);
                if (g + 1 < NG) mmag(g + 1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (khalf) {
#pragma unroll
                for (int r = 0; r < 16; ++r) kp_s[(wq * 16 + r) * 64 + lane] = acc[r];
            }
            __syncthreads();                                     // (waves of the same net: both halves arrive here)
            if (!khalf) {
                const float bias = P[B0(net) + n0 + i];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float part = MLP_NW > 4 ? kp_s[(wq * 16 + r) * 64 + lane] : 0.f;
                    h0_s[(net * 32 + row) * MLP_LD + n0 + i] = tanhf(MLP_NW > 4 ? (acc[r] + part) + bias : acc[r] + bias);
                }
            }
        } else {
            __syncthreads();
        }
    }
    __syncthreads();

    stamp(2);
    // ---- P2: fc1 forward (K = 64: 32 MFMAs per tile, the first four waves)
    {
        const int net = wave >> 1, n0 = (wave & 1) * 32;
        if (wave < 4 && net < nets) {
            const float* W = P + W1(net);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* arow = h0_s + (net * 32 + i) * MLP_LD + 4 * h;
#pragma unroll
            for (int kb = 0; kb < MLP_NH / 8; ++kb) {
                const float4 fa = *reinterpret_cast<const float4*>(arow + 8 * kb);
                const int k = 8 * kb + 4 * h;
                float fb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) fb[q] = W[(k + q) * MLP_NH + n0 + i];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[3], acc, 0, 0, 0);
            }
            const float bias = P[B1(net) + n0 + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                h1_s[(net * 32 + row) * MLP_LD + n0 + i] = tanhf(acc[r] + bias);
            }
        }
    }
    __syncthreads();

    stamp(3);
    // ---- P3a: pdparam (mean / logits) and value, same fmaf order as heads_train_kernel
    const float* lat = h1_s;                                 // policy latent
    const float* vlat = shared ? h1_s : h1_s + 32 * MLP_LD;  // value latent
    for (int q = tid; q < 32 * (nact + 1); q += MLP_NT) {
        const int s = q / (nact + 1), j = q - s * (nact + 1);
        if (j < nact) {
            float acc = 0.f;
            const float* x = lat + s * MLP_LD;
            for (int k = 0; k < MLP_NH; ++k) acc = fmaf(x[k], wpi_s[k * nact + j], acc);
            pi_s[s * 32 + j] = acc + P[a.bpi + j];
        } else {
            float acc = 0.f;
            const float* x = vlat + s * MLP_LD;
            for (int k = 0; k < MLP_NH; ++k) acc = fmaf(x[k], wvf_s[k], acc);
            v_s[s] = acc + P[a.bvf];
        }
    }
    __syncthreads();

    // ---- P3b: per-sample loss and closed-form gradients (ppo2/model.py:57-91)
    if (tid < 32) {
        const int s = tid;
        const long r = row_s[s];
        double st[5] = {0, 0, 0, 0, 0};
        float* pi = pi_s + s * 32;
        float* dpi = dpi_s + s * 32;
        float* dls = dls_s + s * 32;
        if (r >= 0) {
            const float mean = adv_mean, sd = adv_sd;
            const float eps = a.cliprange, ce = a.ent_coef * a.invB;
            const float R = a.returns[r], oldv = a.values[r], oldnlp = a.neglogp[r];
            const float adv = ((R - oldv) - mean) / sd;
            float nlp, H;
            if (a.pd_kind == MRL_PD_CATEGORICAL) {
                const int act = static_cast<const int32_t*>(a.actions)[r];
                float mx = pi[0];
                for (int j = 1; j < nact; ++j) mx = fmaxf(mx, pi[j]);
                float z0 = 0.f;
                for (int j = 0; j < nact; ++j) z0 += expf(pi[j] - mx);
                const float logz = logf(z0);
                nlp = logz - (pi[act] - mx);
                H = 0.f;
                for (int j = 0; j < nact; ++j) {
                    float a0 = pi[j] - mx;
                    H += (expf(a0) / z0) * (logz - a0);
                }
                const float ratio = expf(oldnlp - nlp);
                const float pg1 = -adv * ratio;
                const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
                const float pg2 = -adv * rc;
                float dr = (pg1 >= pg2) ? -adv : ((ratio >= 1.f - eps && ratio <= 1.f + eps) ? -adv : 0.f);
                const float dnlp = dr * (-ratio) * a.invB;
                for (int j = 0; j < nact; ++j) {
                    float a0 = pi[j] - mx;
                    float p = expf(a0) / z0;
                    float logp = a0 - logz;
                    dpi[j] = dnlp * (p - (j == act ? 1.f : 0.f)) + ce * p * (logp + H);
                }
                st[0] = (double)fmaxf(pg1, pg2);
                st[3] = 0.5 * (double)((nlp - oldnlp) * (nlp - oldnlp));
                st[4] = (fabsf(ratio - 1.f) > eps) ? 1.0 : 0.0;
            } else {
                const float* x = static_cast<const float*>(a.actions) + r * nact;
                const float* logstd = P + a.logstd;
                float ssum = 0.f, lsum = 0.f;
                H = 0.f;
                for (int k = 0; k < nact; ++k) {
                    float ls = logstd[k];
                    float u = (x[k] - pi[k]) / expf(ls);
                    ssum += u * u;
                    lsum += ls;
                    H += ls + MRL_MLP_HALF_LOG_2PIE;
                }
                nlp = 0.5f * ssum + MRL_MLP_HALF_LOG_2PI * (float)nact + lsum;
                const float ratio = expf(oldnlp - nlp);
                const float pg1 = -adv * ratio;
                const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
                const float pg2 = -adv * rc;
                float dr = (pg1 >= pg2) ? -adv : ((ratio >= 1.f - eps && ratio <= 1.f + eps) ? -adv : 0.f);
                const float dnlp = dr * (-ratio) * a.invB;
                for (int k = 0; k < nact; ++k) {
                    float sdk = expf(logstd[k]);
                    float u = (x[k] - pi[k]) / sdk;
                    dpi[k] = dnlp * (-(u / sdk));
                    dls[k] = dnlp * (1.f - u * u) - ce;
                }
                st[0] = (double)fmaxf(pg1, pg2);
                st[3] = 0.5 * (double)((nlp - oldnlp) * (nlp - oldnlp));
                st[4] = (fabsf(ratio - 1.f) > eps) ? 1.0 : 0.0;
            }
            const float v = v_s[s];
            const float dvc = fminf(fmaxf(v - oldv, -eps), eps);
            const float vclip = oldv + dvc;
            const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
            float dl = (l1 >= l2) ? (v - R) : ((v - oldv >= -eps && v - oldv <= eps) ? (vclip - R) : 0.f);
            dv_s[s] = a.vf_coef * a.invB * dl;
            st[1] = 0.5 * (double)fmaxf(l1, l2);
            st[2] = (double)H;
        } else {                                    // beyond the minibatch: no contribution
            for (int j = 0; j < nact; ++j) { dpi[j] = 0.f; dls[j] = 0.f; }
            dv_s[s] = 0.f;
        }
        // sum the 32 samples of the tile in lane order (fixed)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            double t = st[j];
            for (int off = 16; off > 0; off >>= 1) t += __shfl_down(t, off, 32);
            if (s == 0) a.spart[(long)blockIdx.x * 5 + j] = t;
        }
    }
    __syncthreads();

    stamp(4);
    // ---- P4: head parameter gradients (slab) + dz of the last hidden layer (LDS)
    {
        const int HPn = 64 * nact + nact;            // pi/w, pi/b
        for (int e = tid; e < HPn; e += MLP_NT) {
            float g = 0.f;
            if (e < 64 * nact) {
                const int k = e / nact, j = e - k * nact;
                for (int s = 0; s < 32; ++s) g = fmaf(lat[s * MLP_LD + k], dpi_s[s * 32 + j], g);
                slab[a.wpi + e] = g;
            } else {
                const int j = e - 64 * nact;
                for (int s = 0; s < 32; ++s) g += dpi_s[s * 32 + j];
                slab[a.bpi + j] = g;
            }
        }
        if (a.logstd >= 0)
            for (int j = tid; j < nact; j += MLP_NT) {
                float g = 0.f;
                for (int s = 0; s < 32; ++s) g += dls_s[s * 32 + j];
                slab[a.logstd + j] = g;
            }
        for (int k = tid; k < 65; k += MLP_NT) {
            float g = 0.f;
            if (k < 64) {
                for (int s = 0; s < 32; ++s) g = fmaf(vlat[s * MLP_LD + k], dv_s[s], g);
                slab[a.wvf + k] = g;
            } else {
                for (int s = 0; s < 32; ++s) g += dv_s[s];
                slab[a.bvf] = g;
            }
        }
        for (int q = tid; q < 32 * 64; q += MLP_NT) {
            const int s = q >> 6, k = q & 63;
            float g = 0.f;
            const float* w = wpi_s + k * nact;
            const float* d = dpi_s + s * 32;
            for (int j = 0; j < nact; ++j) g = fmaf(d[j], w[j], g);
            if (shared) g = fmaf(dv_s[s], wvf_s[k], g);
            const float hv = lat[s * MLP_LD + k];
            dz1_s[s * MLP_LD + k] = g * (1.f - hv * hv);
            if (!shared) {
                const float hvv = vlat[s * MLP_LD + k];
                dz1_s[(32 + s) * MLP_LD + k] = dv_s[s] * wvf_s[k] * (1.f - hvv * hvv);
            }
        }
    }
    __syncthreads();

    stamp(5);
    // ---- P5: fc1 backward.  Tiles per net: dW1 (2x2 tiles, K = 32 samples) and dh0 (2 tiles, K = 64)
    for (int t = wave; t < 6 * nets; t += MLP_NW) {
        const int net = t / 6, q = t % 6;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* h0n = h0_s + net * 32 * MLP_LD;
        const float* dz1n = dz1_s + net * 32 * MLP_LD;
        if (q < 4) {                                         // dW1[m][n] = sum_s h0[s][m] * dz1[s][n]
            const int m0 = (q >> 1) * 32, n0 = (q & 1) * 32;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int s = 2 * kk + h;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h0n[s * MLP_LD + m0 + i], dz1n[s * MLP_LD + n0 + i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[W1(net) + (long)(m0 + row) * MLP_NH + n0 + i] = acc[r];
            }
        } else {                                             // dh0[s][c] = sum_n dz1[s][n] * W1[c][n]
            const int c0 = (q - 4) * 32;
            const float* W = P + W1(net);
            const float* arow = dz1n + i * MLP_LD + 4 * h;
#pragma unroll
            for (int kb = 0; kb < MLP_NH / 8; ++kb) {
                const float4 fa = *reinterpret_cast<const float4*>(arow + 8 * kb);
                const float4 fb = *reinterpret_cast<const float4*>(W + (long)(c0 + i) * MLP_NH + 8 * kb + 4 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float hv = h0n[row * MLP_LD + c0 + i];
                dz0_s[(net * 32 + row) * MLP_LD + c0 + i] = acc[r] * (1.f - hv * hv);
            }
        }
    }
    // bias gradients of fc1: column sums of dz1 (sample order)
    for (int e = tid; e < 64 * nets; e += MLP_NT) {
        const int net = e >> 6, n = e & 63;
        float g = 0.f;
        for (int s = 0; s < 32; ++s) g += dz1_s[(net * 32 + s) * MLP_LD + n];
        slab[B1(net) + n] = g;
    }
    __syncthreads();

    stamp(6);
    // ---- P6: fc0 weight gradient dW0[m][n] = sum_s obs[s][m] * dz0[s][n]  (K = 32 samples per tile)
    {
        const int MT = (K0 + 31) / 32;
        const int ntile = MT * 2 * nets;
        auto ldtile = [&](int t, float (&fa)[16], float (&fbv)[16]) {
            const int net = t / (MT * 2), q = t % (MT * 2);
            const int m0 = (q >> 1) * 32, n0 = (q & 1) * 32;
            const float* dz0n = dz0_s + net * 32 * MLP_LD;
            const int mc = min(m0 + i, KP - 1);              // columns beyond K0 feed rows that are never stored
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int s = 2 * kk + h;
                fa[kk] = obs_s[s * KP + mc];
                fbv[kk] = dz0n[s * MLP_LD + n0 + i];
            }
        };
        auto dotile = [&](int t, const float (&fa)[16], const float (&fbv)[16]) {
            const int net = t / (MT * 2), q = t % (MT * 2);
            const int m0 = (q >> 1) * 32, n0 = (q & 1) * 32;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fbv[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < K0) slab[W0(net) + (long)row * MLP_NH + n0 + i] = acc[r];
            }
        };
        float a0[16], b0v[16], a1[16], b1v[16];
        if (wave < ntile) ldtile(wave, a0, b0v);
        for (int t = wave; t < ntile; t += 2 * MLP_NW) {
            if (t + MLP_NW < ntile) ldtile(t + MLP_NW, a1, b1v);
            dotile(t, a0, b0v);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 * MLP_NW < ntile) ldtile(t + 2 * MLP_NW, a0, b0v);
            if (t + MLP_NW < ntile) dotile(t + MLP_NW, a1, b1v);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int e = tid; e < 64 * nets; e += MLP_NT) {
            const int net = e >> 6, n = e & 63;
            float g = 0.f;
            for (int s = 0; s < 32; ++s) g += dz0_s[(net * 32 + s) * MLP_LD + n];
            slab[B0(net) + n] = g;
        }
    }
    __syncthreads();
    stamp(7);
}

inline size_t mlp_step_lds_bytes(int K0, int nets) {
    const int KP = (K0 + 7) / 8 * 8 + 4;
    size_t floats = (size_t)32 * KP + (size_t)4 * nets * 32 * MLP_LD + 3 * 32 * 32 + 64 + 64 * 32 + 64;
    return floats * 4 + 32 * sizeof(long) + (size_t)4 * 16 * 64 * sizeof(float) + 64;
}

}  // namespace mrl
