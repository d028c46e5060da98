// Act-side forward of the 2 x 64 tanh MLP policy / value networks (common/models.py:74-103 `mlp` behind
// common/policies.py:77-96 step()) as ONE launch (round 6).  The layer-wise path spends four launches of the generic tiled GEMM on
// it -- 1024 x 64 x 376 and 1024 x 64 x 64 products, 16 workgroups each walking K alone: 4 x 22 us of latency for 57 MFLOP, two
// thirds of an env step of the MuJoCo-shaped rollout (profiles/r06m_rollout_mujoco1024_kernel_stats.txt).  Here a workgroup of 128
// threads carries S samples through both layers of both nets: thread (net, column) keeps S accumulators and walks k in order with
// fmaf (fp32 products and sums like the layer-wise path's fp32 matrix instruction; the order of the sums differs in the last bits:
// tests/test_gpu_kernels.py holds the two paths to 2e-6) -- observations and hidden activations sit in LDS (broadcast reads),
// weight columns stream from L2 (256 contiguous bytes per k and net) through two register sets.  No HBM round trip between the
// layers; the heads kernel follows as before.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hip.h"

namespace mrl {

struct MlpActArgs {
    const float* params; const float* obs;
    long w0[2], b0[2], w1[2], b1[2];       // flat offsets; 0 = policy net, 1 = value net (nets == 2)
    float* lat[2];                         // [n][64] each
    int K0, n, nets;
};

// S samples per workgroup; the k range of the first layer is split over KS thread groups (partial sums combined through LDS in a
// fixed order): thread = (k group, net, column), 128 * KS threads.
template <int S, int KS>
__global__ __launch_bounds__(128 * KS) void mlp_act_latent_kernel(MlpActArgs a) {
    constexpr int NH = 64, NT = 128 * KS;
    extern __shared__ __attribute__((aligned(16))) float mlpact_sm[];
    const int K0 = a.K0, KP = (K0 + 3) & ~3;                // row pitch of the observation tile (zero-padded to a multiple of 4)
    float* xs = mlpact_sm;                                  // [S][KP]
    float* hs = xs + S * KP;                                // [2][S][NH]
    float* ps = hs + 2 * S * NH;                            // [KS][2][S][NH] partial sums of the first layer
    const int tid = threadIdx.x, kg = tid >> 7, net = (tid >> 6) & 1, col = tid & 63;
    const long s0 = (long)blockIdx.x * S;
    if (K0 % 4 == 0 && (reinterpret_cast<uintptr_t>(a.obs) & 15) == 0) {
        // the tile's S rows are S * K0 contiguous floats: all 16-byte loads of a thread in flight together (a scalar loop that
        // consumes every load before it issues the next pays one memory round trip per element)
        const float4* src = reinterpret_cast<const float4*>(a.obs + s0 * K0);
        const int nv = S * K0 / 4, nvalid = (int)min((long)nv, (a.n - s0) * (long)(K0 / 4));
        constexpr int MAXV = 8;                             // covers S * K0 <= 32 * NT floats
        float4 v[MAXV];
#pragma unroll
        for (int q = 0; q < MAXV; ++q) {
            const int e = tid + q * NT;
            v[q] = e < nvalid ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < MAXV; ++q) {
            const int e = tid + q * NT;
            if (e < nv) reinterpret_cast<float4*>(xs)[e] = v[q];
        }
        for (int e = tid + MAXV * NT; e < nv; e += NT)      // (wider observations: the rest one by one)
            reinterpret_cast<float4*>(xs)[e] = e < nvalid ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int e = tid; e < S * KP; e += NT) {
            const int s = e / KP, k = e - s * KP;
            xs[e] = (k < K0 && s0 + s < a.n) ? a.obs[(s0 + s) * K0 + k] : 0.f;
        }
    }
    __syncthreads();
    const bool on = net < a.nets;
    // (never index the argument arrays with a runtime value: see mlpstep.hip.h)
    const float* W0 = a.params + (net ? a.w0[1] : a.w0[0]) + col;
    const float* W1 = a.params + (net ? a.w1[1] : a.w1[0]) + col;
    float acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = 0.f;
    if (on) {
        // the weight column streams through two register sets of KU values: the loads of block j + 1 are in flight while block j
        // is multiplied
        constexpr int KU = 16;
        float wa[KU], wb[KU];
        const int nblk_all = K0 / KU;
        const int jb = nblk_all * kg / KS, je = nblk_all * (kg + 1) / KS;      // this k group's blocks
        auto loadw = [&](int j, float (&w)[KU]) {
#pragma unroll
            for (int q = 0; q < KU; ++q) w[q] = W0[(long)(j * KU + q) * NH];
        };
        auto mulw = [&](int j, const float (&w)[KU]) {
#pragma unroll
            for (int q = 0; q < KU; q += 4)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float4 x = *reinterpret_cast<const float4*>(xs + s * KP + j * KU + q);
                    acc[s] = __builtin_fmaf(x.x, w[q], acc[s]);
                    acc[s] = __builtin_fmaf(x.y, w[q + 1], acc[s]);
                    acc[s] = __builtin_fmaf(x.z, w[q + 2], acc[s]);
                    acc[s] = __builtin_fmaf(x.w, w[q + 3], acc[s]);
                }
        };
        if (jb < je) loadw(jb, wa);
        for (int j = jb; j < je; j += 2) {
            if (j + 1 < je) loadw(j + 1, wb);
            mulw(j, wa);
            if (j + 1 < je) {
                if (j + 2 < je) loadw(j + 2, wa);
                mulw(j + 1, wb);
            }
        }
        if (kg == KS - 1)                                    // the k tail rides with the last group
            for (int k = nblk_all * KU; k < K0; ++k) {
                const float w = W0[(long)k * NH];
#pragma unroll
                for (int s = 0; s < S; ++s) acc[s] = __builtin_fmaf(xs[s * KP + k], w, acc[s]);
            }
#pragma unroll
        for (int s = 0; s < S; ++s) ps[((kg * 2 + net) * S + s) * NH + col] = acc[s];
    }
    __syncthreads();
    if (on && kg == 0) {
        const float b = a.params[(net ? a.b0[1] : a.b0[0]) + col];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float t = ps[(net * S + s) * NH + col];
#pragma unroll
            for (int g = 1; g < KS; ++g) t += ps[((g * 2 + net) * S + s) * NH + col];          // fixed order
            hs[(net * S + s) * NH + col] = tanhf(t + b);
        }
    }
    __syncthreads();
    if (on && kg == 0) {
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = 0.f;
        float w1r[NH];
#pragma unroll
        for (int k = 0; k < NH; ++k) w1r[k] = W1[(long)k * NH];
#pragma unroll
        for (int k = 0; k < NH; k += 4) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float4 x = *reinterpret_cast<const float4*>(hs + (net * S + s) * NH + k);
                acc[s] = __builtin_fmaf(x.x, w1r[k], acc[s]);
                acc[s] = __builtin_fmaf(x.y, w1r[k + 1], acc[s]);
                acc[s] = __builtin_fmaf(x.z, w1r[k + 2], acc[s]);
                acc[s] = __builtin_fmaf(x.w, w1r[k + 3], acc[s]);
            }
        }
        const float b = a.params[(net ? a.b1[1] : a.b1[0]) + col];
        float* out = net ? a.lat[1] : a.lat[0];
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s0 + s < a.n) out[(s0 + s) * NH + col] = tanhf(acc[s] + b);
    }
}

inline size_t mlp_act_lds_bytes(int K0, int S, int KS) { return (size_t)(S * ((K0 + 3) & ~3) + 2 * S * 64 + KS * 2 * S * 64) * sizeof(float); }

inline hipError_t launch_mlp_act_latent(const MlpActArgs& a, hipStream_t stream) {
    constexpr int S = 4, KS = 2;
    const size_t lds = mlp_act_lds_bytes(a.K0, S, KS);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL((mlp_act_latent_kernel<S, KS>), dim3((unsigned)((a.n + S - 1) / S)), dim3(128 * KS), lds, stream, a);
    return hipGetLastError();
}

}  // namespace mrl
