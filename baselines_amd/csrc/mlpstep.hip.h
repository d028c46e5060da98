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
    // slice != 0 (two separate nets only): the grid has TWO workgroups per tile -- workgroup 2 t carries the policy net, its
    // head and the policy part of the loss through the step, workgroup 2 t + 1 the value net; each writes its own part of
    // slab t and its own entries of the 5 statistics (they are disjoint), and only the policy workgroup needs the advantage
    // statistics.  A tile's serial chain is cut in half and a 4096-sample minibatch fills 256 CUs instead of 128.
    int slice;
    float* part;           // [ntiles][P]
    double* spart;         // [ntiles][5]
    long long* dbg;        // optional [8] phase timestamps (shader clock) of workgroup dbg_block, nullptr normally
    int dbg_block;
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
    const int nact = a.nact;
    const bool shared = a.nets == 1;                        // one latent feeds both heads
    const bool sl = a.slice != 0;                           // this workgroup carries ONE of the two nets
    const int tile = sl ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int nb = sl ? (int)(blockIdx.x & 1) : 0;          // first net of this workgroup (0: policy, 1: value)
    const int nets = sl ? 1 : a.nets;                       // nets this workgroup carries (local index 0 .. nets-1)
    const bool do_pi = !sl || nb == 0, do_vf = !sl || nb == 1;
    const int s0 = tile * 32;
    const float* P = a.params;
    float* slab = a.part + (long)tile * a.P;
    // NOTE: never index a.w0[] etc. with a runtime value -- a dynamically indexed kernel-argument array is
    // spilled to memory and re-loaded (global_load + s_waitcnt vmcnt(0)) in front of every use.
    // (n = LOCAL net index; nb shifts it to the net whose parameters are meant)
    auto W0 = [&](int n) { return (n + nb) ? a.w0[1] : a.w0[0]; };
    auto B0 = [&](int n) { return (n + nb) ? a.b0[1] : a.b0[0]; };
    auto W1 = [&](int n) { return (n + nb) ? a.w1[1] : a.w1[0]; };
    auto B1 = [&](int n) { return (n + nb) ? a.b1[1] : a.b1[0]; };
    auto stamp = [&](int k) { if (a.dbg && (int)blockIdx.x == a.dbg_block && tid == 0) a.dbg[k] = (long long)__builtin_readcyclecounter(); };
    const int anets = nets;                                 // LDS is carved for the nets THIS workgroup carries
    stamp(0);

    // ---- LDS carve
    float* obs_s = sm;                                   // [32][KP]
    float* h0_s = obs_s + 32 * KP;                       // [nets][32][MLP_LD]
    float* h1_s = h0_s + anets * 32 * MLP_LD;
    float* dz1_s = h1_s + anets * 32 * MLP_LD;
    float* dz0_s = dz1_s + anets * 32 * MLP_LD;
    float* pi_s = dz0_s + anets * 32 * MLP_LD;           // [32][32] pdparam (mean / logits)
    float* dpi_s = pi_s + 32 * 32;                       // [32][32]
    float* dls_s = dpi_s + 32 * 32;                      // [32][32]
    float* v_s = dls_s + 32 * 32;                        // [32]
    float* dv_s = v_s + 32;                              // [32]
    float* wpi_s = dv_s + 32;                            // [64][nact] policy head weights
    float* wvf_s = wpi_s + 64 * 32;                      // [64] value head weights
    long* row_s = reinterpret_cast<long*>(wvf_s + 64);   // [32] storage rows (-1: beyond the minibatch)
    // [waves][16][64] accumulator exchange of the two forward layers (k-range partial sums; sharing out the tanh epilogue).
    // A workgroup that carries two nets borrows the dz1 | dz0 area for it (first written in P4), one net has its own 32 KB.
    float* kp_s = nets == 1 ? reinterpret_cast<float*>(row_s + 32) : dz1_s;

    // ---- weight fragments of the two forward layers: requested FIRST.  The parameters were rewritten by the optimizer since
    //      the last step, so every XCD fetches them from the fabric again; asked for here, that trip overlaps with P0's
    //      index / gather chains instead of standing in front of the MFMAs of P1 / P2 (P1: 9.5 -> ... us).
    const int p1_NTL = 2 * nets, p1_KS = MLP_NW / p1_NTL;
    const int p1_wq = wave % p1_NTL, p1_kr = wave / p1_NTL;         // output tile, k range of this wave
    const int p1_net = p1_wq >> 1, p1_n0 = (p1_wq & 1) * 32;
    const int p1_NGT = (KB0 + 3) / 4;                               // groups of 4 k blocks; range p1_kr of p1_KS
    const int p1_gb = (p1_NGT * p1_kr + p1_KS - 1) / p1_KS, p1_NG = (p1_NGT * (p1_kr + 1) + p1_KS - 1) / p1_KS;
    const float* p1_wcol = P + W0(p1_net) + p1_n0 + i;
    auto loadg = [&](int g, float (&dst)[16]) {                     // unconditional loads with a clamped row, zeroed by select
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 8 * (4 * g + u) + 4 * h + q;
                const float w = p1_wcol[(long)min(k, K0 - 1) * MLP_NH];
                dst[4 * u + q] = k < K0 ? w : 0.f;
            }
    };
    const bool p1_pre = p1_NG - p1_gb <= 3;                         // short k range (3 groups at K0 = 376 with 8 waves on one net)
    float fb0[16], fb1[16], fb2[16];
    if (p1_pre) {
        if (p1_gb < p1_NG) loadg(p1_gb, fb0);
        if (p1_gb + 1 < p1_NG) loadg(p1_gb + 1, fb1);
        if (p1_gb + 2 < p1_NG) loadg(p1_gb + 2, fb2);
    }
    const float p1_bias = P[B0(p1_net) + p1_n0 + i];            // (every wave finishes a share of its tile: tanh is not cheap)
    const float p2t_bias = P[B1(p1_net) + p1_n0 + i];           // same tile roles in the epilogue of P2
    const int hq0 = tid;                                        // P3a: head bias of this thread's first output
    const float hb0 = hq0 < 32 * (nact + 1) ? P[(hq0 % (nact + 1)) < nact ? a.bpi + (hq0 % (nact + 1)) : a.bvf] : 0.f;
    const int p2_net = wave >> 1, p2_n0 = (wave & 1) * 32;
    const bool p2_on = wave < 4 && p2_net < nets;
    const float p2_bias = p2_on ? P[B1(p2_net) + p2_n0 + i] : 0.f;
    float w1f[32];                                                   // fc1 weight fragments of this wave's output tile
    if (p2_on) {
        const float* W = P + W1(p2_net);
#pragma unroll
        for (int kb = 0; kb < MLP_NH / 8; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) w1f[4 * kb + q] = W[(8 * kb + 4 * h + q) * MLP_NH + p2_n0 + i];
    }

    // ---- P0: rows, advantage statistics of the whole minibatch (model.py:136-139; f64 accumulation, fixed order), observation
    //      tile, small weights.  One wave per SIMD: every memory latency is exposed, so the two dependent chains (index ->
    //      returns / values of the Bstat samples; index -> observation row) are walked TOGETHER: all index loads, then all
    //      gathers, then the reductions and LDS writes (round 3; one chain after the other cost 16 of the step's 53 us).
    auto tile_row = [&](int sidx) -> long {                 // storage row of sample sidx of this tile (-1: beyond the minibatch)
        const int b = s0 + sidx;
        if (b >= a.B) return -1;
        return a.tile_idx ? envmajor_to_row(a.tile_idx[b], a.T, a.N) : (a.srow ? (long)a.srow[b] : (long)b);
    };
    if (tid < 32) row_s[tid] = tile_row(tid);
    float adv_mean = 0.f, adv_sd = 1.f;
    const bool own_stats = do_pi && !a.advstat;             // (a value workgroup of a sliced launch never needs the advantage)
    {
        const int nv = 32 * (KP / 4);
        const int nob = (nv + 8 * MLP_NT - 1) / (8 * MLP_NT);          // batches of 8 float4 per thread (1 at K0 = 376)
        const int nsb = own_stats ? (a.Bstat + 8 * MLP_NT - 1) / (8 * MLP_NT) : 0;
        double s1 = 0.0, s2 = 0.0;
        for (int it = 0; it < max(nob, nsb); ++it) {
            const int b0 = tid + it * 8 * MLP_NT, e0 = tid + it * 8 * MLP_NT;
            long sr[8], orow[8];
            float rv[8], vv[8];
            float4 v[8];
            // round 1: indices
            if (it < nsb) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = min(b0 + u * MLP_NT, a.Bstat - 1);
                    sr[u] = a.stat_idx ? (long)a.stat_idx[b] : (long)b;
                }
            }
            if (it < nob) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * MLP_NT;
                    orow[u] = e < nv ? tile_row(e / (KP / 4)) : -1;
                }
            }
            // round 2: gathers
            if (it < nsb) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (a.stat_idx) sr[u] = envmajor_to_row(sr[u], a.T, a.N);
                    rv[u] = a.stat_ret[sr[u]];
                    vv[u] = a.stat_val[sr[u]];
                }
            }
            if (it < nob) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * MLP_NT;
                    v[u] = f4zero();
                    if (e < nv) {
                        const int c4 = (e % (KP / 4)) * 4;
                        const long r = orow[u];
                        if (r >= 0 && c4 + 3 < K0) v[u] = *reinterpret_cast<const float4*>(a.obs + r * K0 + c4);
                        else if (r >= 0 && c4 < K0) v[u] = load_partial(a.obs + r * K0 + c4, K0 - c4);
                    }
                }
            }
            if (it == 0) {      // small weights -> LDS (coalesced): pi head, value head
                if (do_pi)
                    for (int e = tid; e < 64 * nact; e += MLP_NT) wpi_s[e] = P[a.wpi + e];
                if (do_vf)
                    for (int e = tid; e < 64; e += MLP_NT) wvf_s[e] = P[a.wvf + e];
            }
            if (it < nsb) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (b0 + u * MLP_NT < a.Bstat) {
                        const float x = __fsub_rn(rv[u], vv[u]);
                        s1 += (double)x;
                        s2 += (double)x * (double)x;
                    }
            }
            if (it < nob) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * MLP_NT;
                    if (e < nv) {
                        const int sidx = e / (KP / 4), c4 = (e % (KP / 4)) * 4;
                        *reinterpret_cast<float4*>(obs_s + sidx * KP + c4) = v[u];
                    }
                }
            }
        }
        if (own_stats) {
            double* red = reinterpret_cast<double*>(pi_s);       // LDS scratch; pi_s is first written in P3a
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
        } else if (do_pi) {
            adv_mean = a.advstat[0];
            adv_sd = a.advstat[1] + 1e-8f;
        }
    }
    __syncthreads();

    // per-sample scalars of the loss (P3b) are fetched NOW: their latency hides behind the two forward layers
    constexpr int LPS = MLP_NT / 32;                         // lanes per sample in P3b: 16 | 8
    const int ls_s = tid / LPS, ls_q = tid % LPS;
    const long ls_r = row_s[ls_s];
    float pf_R = 0.f, pf_oldv = 0.f, pf_oldnlp = 0.f, pf_x[2] = {0.f, 0.f}, pf_ls[2] = {0.f, 0.f};
    int pf_act = 0;
    if (ls_r >= 0) {
        pf_R = a.returns[ls_r]; pf_oldv = a.values[ls_r]; pf_oldnlp = a.neglogp[ls_r];
        if (do_pi) {
            if (a.pd_kind == MRL_PD_CATEGORICAL) {
                pf_act = static_cast<const int32_t*>(a.actions)[ls_r];
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = ls_q + u * LPS;
                    if (k < nact) {
                        pf_x[u] = static_cast<const float*>(a.actions)[ls_r * nact + k];
                        pf_ls[u] = P[a.logstd + k];
                    }
                }
            }
        }
    }
    stamp(1);
    // ---- P1: fc0 forward.  wave -> (k range, net, 32-column tile): the 2 * nets output tiles are shared out over the waves,
    //      KS = waves / (2 nets) of them splitting the k range of one tile (8 waves: 2 ranges for two nets, 4 for one)
    {
        const int NTL = p1_NTL, KS = p1_KS, wq = p1_wq, khalf = p1_kr, net = p1_net, n0 = p1_n0, gb = p1_gb, NG = p1_NG;
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* arow = obs_s + i * KP + 4 * h;
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
            if (p1_pre) {
                if (gb < NG) mmag(gb, fb0);
                if (gb + 1 < NG) mmag(gb + 1, fb1);
                if (gb + 2 < NG) mmag(gb + 2, fb2);
            } else {
                // long k range: the 16 loads of the NEXT group of 4 k-blocks are issued before the 16 MFMAs of the current one
                if (gb < NG) loadg(gb, fb0);
                for (int g = gb; g < NG; g += 2) {
                    if (g + 1 < NG) loadg(g + 1, fb1);
                    mmag(g, fb0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 2 < NG) loadg(g + 2, fb0);
                    if (g + 1 < NG) mmag(g + 1, fb1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the KS waves of a tile exchange their partial accumulators through LDS; wave `khalf` then finishes 16 / KS of the
            // 16 accumulator registers (partials summed in k-range order, bias, tanh): the 16 tanh evaluations per lane that
            // one wave used to walk through (4 of P1's 8 us) are shared out
            if (KS > 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) kp_s[((khalf * NTL + wq) * 16 + r) * 64 + lane] = acc[r];
            }
            __syncthreads();                                     // (every wave arrives here)
            {
                const float bias = p1_bias;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r * KS / 16 != khalf) continue;          // wave-uniform
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    float v = KS > 1 ? kp_s[(wq * 16 + r) * 64 + lane] : acc[r];
                    for (int q = 1; q < KS; ++q) v += kp_s[((q * NTL + wq) * 16 + r) * 64 + lane];           // fixed order
                    h0_s[(net * 32 + row) * MLP_LD + n0 + i] = tanhf(v + bias);
                }
            }
        }
    }
    __syncthreads();

    stamp(2);
    // ---- P2: fc1 forward (K = 64: 32 MFMAs per tile, the first four waves)
    {
        const int net = p2_net, n0 = p2_n0;
        if (p2_on) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* arow = h0_s + (net * 32 + i) * MLP_LD + 4 * h;
#pragma unroll
            for (int kb = 0; kb < MLP_NH / 8; ++kb) {
                const float4 fa = *reinterpret_cast<const float4*>(arow + 8 * kb);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, w1f[4 * kb + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, w1f[4 * kb + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, w1f[4 * kb + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, w1f[4 * kb + 3], acc, 0, 0, 0);
            }
            if (p1_KS > 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) kp_s[(wave * 16 + r) * 64 + lane] = acc[r];       // wave == tile index here
            } else {
                const float bias = p2_bias;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    h1_s[(net * 32 + row) * MLP_LD + n0 + i] = tanhf(acc[r] + bias);
                }
            }
        }
        if (p1_KS > 1) {
            // bias + tanh of the tile shared out like in P1: wave (tile p1_wq, part p1_kr) finishes 16 / KS registers
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r * p1_KS / 16 != p1_kr) continue;
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                h1_s[(p1_net * 32 + row) * MLP_LD + p1_n0 + i] = tanhf(kp_s[(p1_wq * 16 + r) * 64 + lane] + p2t_bias);
            }
        }
    }
    __syncthreads();

    stamp(3);
    // ---- P3a: pdparam (mean / logits) and value, same fmaf order as heads_train_kernel
    const float* lat = h1_s;                                 // policy latent
    const float* vlat = (shared || sl) ? h1_s : h1_s + 32 * MLP_LD;  // value latent (sliced: this workgroup's only net)
    for (int q = tid; q < 32 * (nact + 1); q += MLP_NT) {
        const int s = q / (nact + 1), j = q - s * (nact + 1);
        if (j < nact ? !do_pi : !do_vf) continue;
        const float hb = q == hq0 ? hb0 : P[j < nact ? a.bpi + j : a.bvf];      // head bias: asked for at kernel start
        if (j < nact) {
            float acc = 0.f;
            const float* x = lat + s * MLP_LD;
            for (int k = 0; k < MLP_NH; ++k) acc = fmaf(x[k], wpi_s[k * nact + j], acc);
            pi_s[s * 32 + j] = acc + hb;
        } else {
            float acc = 0.f;
            const float* x = vlat + s * MLP_LD;
            for (int k = 0; k < MLP_NH; ++k) acc = fmaf(x[k], wvf_s[k], acc);
            v_s[s] = acc + hb;
        }
    }
    __syncthreads();

    // ---- P3b: per-sample loss and closed-form gradients (ppo2/model.py:57-91).  MLP_NT / 32 consecutive
    //      lanes share a sample and split its action dimensions (round 3; one lane per sample walked 2 x nact exp / divide
    //      chains: 8 of the step's 53 us); sums over the action dimensions are butterfly sums inside the lane group.
    {
        const int s = ls_s, q = ls_q;
        const long r = ls_r;
        double st[5] = {0, 0, 0, 0, 0};
        float* pi = pi_s + s * 32;
        float* dpi = dpi_s + s * 32;
        float* dls = dls_s + s * 32;
        auto gsum = [&](float v) {
#pragma unroll
            for (int off = LPS / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, LPS);
            return v;
        };
        auto gmax = [&](float v) {
#pragma unroll
            for (int off = LPS / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, LPS));
            return v;
        };
        if (r >= 0) {
            const float mean = adv_mean, sd = adv_sd;
            const float eps = a.cliprange, ce = a.ent_coef * a.invB;
            const float R = pf_R, oldv = pf_oldv, oldnlp = pf_oldnlp;
            const float adv = ((R - oldv) - mean) / sd;
            float H = 0.f;
            if (!do_pi) {
                // value workgroup of a sliced launch
            } else if (a.pd_kind == MRL_PD_CATEGORICAL) {
                const int act = pf_act;
                float mx = -INFINITY;
                for (int j = q; j < nact; j += LPS) mx = fmaxf(mx, pi[j]);
                mx = gmax(mx);
                float z0 = 0.f;
                for (int j = q; j < nact; j += LPS) z0 += expf(pi[j] - mx);
                z0 = gsum(z0);
                const float logz = logf(z0);
                const float nlp = logz - (pi[act] - mx);
                for (int j = q; j < nact; j += LPS) {
                    const float a0 = pi[j] - mx;
                    H += (expf(a0) / z0) * (logz - a0);
                }
                H = gsum(H);
                const float ratio = expf(oldnlp - nlp);
                const float pg1 = -adv * ratio;
                const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
                const float pg2 = -adv * rc;
                const float dr = (pg1 >= pg2) ? -adv : ((ratio >= 1.f - eps && ratio <= 1.f + eps) ? -adv : 0.f);
                const float dnlp = dr * (-ratio) * a.invB;
                for (int j = q; j < nact; j += LPS) {
                    const float a0 = pi[j] - mx;
                    const float p = expf(a0) / z0;
                    const float logp = a0 - logz;
                    dpi[j] = dnlp * (p - (j == act ? 1.f : 0.f)) + ce * p * (logp + H);
                }
                st[0] = (double)fmaxf(pg1, pg2);
                st[3] = 0.5 * (double)((nlp - oldnlp) * (nlp - oldnlp));
                st[4] = (fabsf(ratio - 1.f) > eps) ? 1.0 : 0.0;
            } else {
                // nact <= 32 <= 2 LPS when LPS = 16; the 8-lane form walks up to 4 dimensions per lane (the last two re-read)
                const float* x = static_cast<const float*>(a.actions) + r * nact;
                const float* logstd = P + a.logstd;
                auto xk = [&](int k) { const int u = (k - q) / LPS; return u < 2 ? pf_x[u & 1] : x[k]; };
                auto lsk = [&](int k) { const int u = (k - q) / LPS; return u < 2 ? pf_ls[u & 1] : logstd[k]; };
                float ssum = 0.f, lsum = 0.f;
                for (int k = q; k < nact; k += LPS) {
                    const float ls = lsk(k);
                    const float u = (xk(k) - pi[k]) / expf(ls);
                    ssum += u * u;
                    lsum += ls;
                    H += ls + MRL_MLP_HALF_LOG_2PIE;
                }
                ssum = gsum(ssum); lsum = gsum(lsum); H = gsum(H);
                const float nlp = 0.5f * ssum + MRL_MLP_HALF_LOG_2PI * (float)nact + lsum;
                const float ratio = expf(oldnlp - nlp);
                const float pg1 = -adv * ratio;
                const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
                const float pg2 = -adv * rc;
                const float dr = (pg1 >= pg2) ? -adv : ((ratio >= 1.f - eps && ratio <= 1.f + eps) ? -adv : 0.f);
                const float dnlp = dr * (-ratio) * a.invB;
                for (int k = q; k < nact; k += LPS) {
                    const float sdk = expf(lsk(k));
                    const float u = (xk(k) - pi[k]) / sdk;
                    dpi[k] = dnlp * (-(u / sdk));
                    dls[k] = dnlp * (1.f - u * u) - ce;
                }
                st[0] = (double)fmaxf(pg1, pg2);
                st[3] = 0.5 * (double)((nlp - oldnlp) * (nlp - oldnlp));
                st[4] = (fabsf(ratio - 1.f) > eps) ? 1.0 : 0.0;
            }
            if (do_vf && q == 0) {
                const float v = v_s[s];
                const float dvc = fminf(fmaxf(v - oldv, -eps), eps);
                const float vclip = oldv + dvc;
                const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
                const float dl = (l1 >= l2) ? (v - R) : ((v - oldv >= -eps && v - oldv <= eps) ? (vclip - R) : 0.f);
                dv_s[s] = a.vf_coef * a.invB * dl;
                st[1] = 0.5 * (double)fmaxf(l1, l2);
            }
            st[2] = (double)H;
        } else {                                    // beyond the minibatch: no contribution
            for (int j = q; j < nact; j += LPS) { dpi[j] = 0.f; dls[j] = 0.f; }
            if (q == 0) dv_s[s] = 0.f;
        }
        // the tile's 32 samples are summed in sample order by the first 32 lanes (fixed); a sliced workgroup writes the
        // statistics of its own part of the loss
        double* st_s = reinterpret_cast<double*>(kp_s);      // [32][5]; kp_s is free since the end of P1
        if (q == 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j) st_s[s * 5 + j] = st[j];
        }
        __syncthreads();
        if (tid < 32) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                double t = st_s[tid * 5 + j];
                for (int off = 16; off > 0; off >>= 1) t += __shfl_down(t, off, 32);
                if (tid == 0 && (j == 1 ? do_vf : do_pi)) a.spart[(long)tile * 5 + j] = t;
            }
        }
    }
    __syncthreads();

    stamp(4);
    // ---- P4: head parameter gradients (slab) + dz of the last hidden layer (LDS)
    {
        const int HPn = do_pi ? 64 * nact + nact : 0;            // pi/w, pi/b
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
        if (a.logstd >= 0 && do_pi)
            for (int j = tid; j < nact; j += MLP_NT) {
                float g = 0.f;
                for (int s = 0; s < 32; ++s) g += dls_s[s * 32 + j];
                slab[a.logstd + j] = g;
            }
        for (int k = tid; k < (do_vf ? 65 : 0); k += MLP_NT) {
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
            if (do_pi) {
                float g = 0.f;
                const float* w = wpi_s + k * nact;
                const float* d = dpi_s + s * 32;
                for (int j = 0; j < nact; ++j) g = fmaf(d[j], w[j], g);
                if (shared) g = fmaf(dv_s[s], wvf_s[k], g);
                const float hv = lat[s * MLP_LD + k];
                dz1_s[s * MLP_LD + k] = g * (1.f - hv * hv);
            }
            if (!shared && do_vf) {                          // the value net's rows: local net 1, or 0 in a sliced workgroup
                const float hvv = vlat[s * MLP_LD + k];
                dz1_s[((sl ? 0 : 32) + s) * MLP_LD + k] = dv_s[s] * wvf_s[k] * (1.f - hvv * hvv);
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

// nets = nets carried by ONE workgroup (1 in a sliced launch)
// The advantage statistics of `gridDim.x` minibatches (idx [gridDim.x][B] env-major indices) in ONE launch: out[mb] = (mean,
// std) of returns - values over the minibatch -- exactly the sums, in exactly the order, every policy workgroup of
// mlp_step_kernel<NT> otherwise takes for itself (thread-strided batches of 8, f64 accumulation, mlp_block_sum), so a step fed
// from here is bit-identical to one that computes them on its own.  One launch per epoch instead of 128 redundant gathers
// of 4096 samples in front of every step (13 of the step kernel's 48 us).
template <int NT>
__global__ __launch_bounds__(NT) void mlp_advstat_kernel(const float* __restrict__ ret, const float* __restrict__ val,
                                                         const int64_t* __restrict__ idx, int B, int T, int N,
                                                         float* __restrict__ out) {
    constexpr int NW = NT / 64;
    __shared__ double red[2 * NW + 2];
    const int tid = threadIdx.x;
    const int64_t* my = idx + (long)blockIdx.x * B;
    double s1 = 0.0, s2 = 0.0;
    const int nsb = (B + 8 * NT - 1) / (8 * NT);
    for (int it = 0; it < nsb; ++it) {
        const int b0 = tid + it * 8 * NT;
        long sr[8];
        float rv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) sr[u] = (long)my[min(b0 + u * NT, B - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            sr[u] = envmajor_to_row(sr[u], T, N);
            rv[u] = ret[sr[u]];
            vv[u] = val[sr[u]];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (b0 + u * NT < B) {
                const float x = __fsub_rn(rv[u], vv[u]);
                s1 += (double)x;
                s2 += (double)x * (double)x;
            }
    }
    const double t1 = mlp_block_sum<NW>(s1, red);
    const double t2 = mlp_block_sum<NW>(s2, red + NW);
    if (tid == 0) {
        const double mean = t1 / B;
        double var = t2 / B - mean * mean;
        if (var < 0) var = 0;
        out[2 * blockIdx.x] = (float)mean;
        out[2 * blockIdx.x + 1] = (float)sqrt(var);
    }
}

inline size_t mlp_step_lds_bytes(int K0, int nets) {
    const int KP = (K0 + 7) / 8 * 8 + 4;
    size_t floats = (size_t)32 * KP + (size_t)4 * nets * 32 * MLP_LD + 3 * 32 * 32 + 64 + 64 * 32 + 64;
    return floats * 4 + 32 * sizeof(long) + (nets == 1 ? (size_t)8 * 16 * 64 * sizeof(float) : 0) + 64;
}

}  // namespace mrl
