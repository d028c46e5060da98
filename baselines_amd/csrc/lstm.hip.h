// Recurrent cell of the reference's `lstm` / `cnn_lstm` policies on gfx950: forward and backward SCANS over the nsteps of
// a rollout (a2c/utils.py:81-102 `lstm(xs, ms, s, scope, nh)`; common/models.py:132-210), SURVEY.md 8 f4.
//
//     c = c*(1-m);  h = h*(1-m);  z = x@wx + h@wh + b;  i,f,o,u = split(z);  i,f,o = sigmoid;  u = tanh
//     c = f*c + i*u;  h = o*tanh(c)
//
// The input projection x@wx for ALL steps is one GEMM outside (model.hip, existing engines); what is inherently
// sequential is h@wh (nh x 4nh, 256 KB at nh = 128) once per step.  A workgroup of 4*nh threads keeps wh in REGISTERS for
// its whole life (thread j holds column j: nh VGPRs -- this is what 512 VGPRs per SIMD lane buy) and walks E
// environments through the T steps together: per step E*nh fmaf per thread against h broadcast from LDS, three barriers,
// no global traffic besides the streamed gate pre-activations.  Plain fp32 fmaf chains in k order (bitwise the chain of
// a scalar implementation).  The backward scan holds wh in the transposed ownership (thread (k, q) holds row k, columns
// q*nh..q*nh+nh) for dh_prev = dz @ wh^T and writes the gate gradients dz [B][4nh]; the weight gradients
// dwx = X^T dz, dwh = Hm^T dz, db and the input gradient dz @ wx^T are GEMMs over all steps at once (model.hip).
// Everything the backward pass needs is stored by the forward scan (gates, masked c/h, tanh(c): 7*nh floats per sample,
// 3.6 KB at nh = 128 next to NatureCNN's 173 KB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hip.h"

namespace mrl {

struct LstmFwdArgs {
    const float* zx;        // [B][4nh] x@wx (no bias), sample b = e*T + t (env-major)
    const float* wh;        // [nh][4nh]
    const float* bias;      // [4nh]
    const float* s0;        // [nenv][2nh]  (c | h), nullptr: zeros
    const uint8_t* mask;    // done flag entering step t of sample b: mask[srow ? srow[b] : b]
    const int32_t* srow;
    int nenv, T;
    float *gates, *cm, *hm, *tc, *hout;   // [B][4nh], [B][nh] x4; gates/cm/hm/tc may be nullptr (act side)
    float* s_out;           // [nenv][2nh] final state, may be nullptr
};

__device__ __forceinline__ float lstm_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

template <int NH, int E>
__global__ __launch_bounds__(4 * NH) void lstm_fwd_kernel(LstmFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float h_s[E][NH];
    __shared__ float c_s[E][NH];
    __shared__ float g_s[E][4 * NH];
    const int j = threadIdx.x;
    float w[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) w[k] = a.wh[k * 4 * NH + j];
    const float bj = a.bias[j];
    const int T = a.T;
    for (int g0 = blockIdx.x * E; g0 < a.nenv; g0 += gridDim.x * E) {
        for (int q = j; q < E * NH; q += 4 * NH) {
            const int e = q / NH, k = q - e * NH, env = g0 + e;
            const bool ok = env < a.nenv && a.s0;
            c_s[e][k] = ok ? a.s0[(long)env * 2 * NH + k] : 0.f;
            h_s[e][k] = ok ? a.s0[(long)env * 2 * NH + NH + k] : 0.f;
        }
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            float zxv[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int env = min(g0 + e, a.nenv - 1);
                zxv[e] = a.zx[((long)env * T + t) * 4 * NH + j];            // in flight during the mask phase
            }
            for (int q = j; q < E * NH; q += 4 * NH) {
                const int e = q / NH, k = q - e * NH, env = g0 + e;
                if (env < a.nenv) {
                    const long b = (long)env * T + t;
                    const float keep = a.mask[a.srow ? a.srow[b] : b] ? 0.f : 1.f;
                    const float cv = c_s[e][k] * keep, hv = h_s[e][k] * keep;
                    c_s[e][k] = cv;
                    h_s[e][k] = hv;
                    if (a.cm) { a.cm[b * NH + k] = cv; a.hm[b * NH + k] = hv; }
                }
            }
            __syncthreads();
            float acc[E];
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < NH; k4 += 4) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float4 hv = *reinterpret_cast<const float4*>(&h_s[e][k4]);      // broadcast read
                    acc[e] = fmaf(hv.x, w[k4], acc[e]);
                    acc[e] = fmaf(hv.y, w[k4 + 1], acc[e]);
                    acc[e] = fmaf(hv.z, w[k4 + 2], acc[e]);
                    acc[e] = fmaf(hv.w, w[k4 + 3], acc[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float z = (zxv[e] + acc[e]) + bj;                      // (x@wx + h@wh) + b, the reference's order
                const float gv = j < 3 * NH ? lstm_sigmoid(z) : tanhf(z);
                g_s[e][j] = gv;
                if (a.gates && g0 + e < a.nenv) a.gates[((long)(g0 + e) * T + t) * 4 * NH + j] = gv;
            }
            __syncthreads();
            for (int q = j; q < E * NH; q += 4 * NH) {
                const int e = q / NH, k = q - e * NH, env = g0 + e;
                const float iv = g_s[e][k], fv = g_s[e][NH + k], ov = g_s[e][2 * NH + k], uv = g_s[e][3 * NH + k];
                const float c = fv * c_s[e][k] + iv * uv;
                const float tcv = tanhf(c);
                const float h = ov * tcv;
                c_s[e][k] = c;
                h_s[e][k] = h;
                if (env < a.nenv) {
                    const long b = (long)env * T + t;
                    if (a.tc) a.tc[b * NH + k] = tcv;
                    a.hout[b * NH + k] = h;
                }
            }
            __syncthreads();
        }
        if (a.s_out)
            for (int q = j; q < E * NH; q += 4 * NH) {
                const int e = q / NH, k = q - e * NH, env = g0 + e;
                if (env < a.nenv) {
                    a.s_out[(long)env * 2 * NH + k] = c_s[e][k];
                    a.s_out[(long)env * 2 * NH + NH + k] = h_s[e][k];
                }
            }
        __syncthreads();
    }
}

struct LstmBwdArgs {
    const float* dhout;     // [B][nh] dL/dh_t from the heads
    const float* wh;        // [nh][4nh]
    const float *gates, *cm, *tc;
    const uint8_t* mask; const int32_t* srow;
    int nenv, T;
    float* dzg;             // [B][4nh] gradient w.r.t. the gate pre-activations
};

template <int NH, int E>
__global__ __launch_bounds__(4 * NH) void lstm_bwd_kernel(LstmBwdArgs a) {
    __shared__ float dh_s[E][NH];
    __shared__ float dc_s[E][NH];
    __shared__ __attribute__((aligned(16))) float dz_s[E][4 * NH];
    __shared__ float p_s[E][4][NH];
    const int tid = threadIdx.x;
    const int k = tid % NH, q4 = tid / NH;
    float w[NH];                                  // wh[k][q4*NH + jj]
#pragma unroll
    for (int jj = 0; jj < NH; ++jj) w[jj] = a.wh[(long)k * 4 * NH + q4 * NH + jj];
    const int T = a.T;
    for (int g0 = blockIdx.x * E; g0 < a.nenv; g0 += gridDim.x * E) {
        for (int q = tid; q < E * NH; q += 4 * NH) { dh_s[q / NH][q % NH] = 0.f; dc_s[q / NH][q % NH] = 0.f; }
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            for (int q = tid; q < E * NH; q += 4 * NH) {
                const int e = q / NH, kk = q - e * NH, env = g0 + e;
                float dzi = 0.f, dzf = 0.f, dzo = 0.f, dzu = 0.f, dcm = 0.f;
                if (env < a.nenv) {
                    const long b = (long)env * T + t;
                    const float* gr = a.gates + b * 4 * NH;
                    const float iv = gr[kk], fv = gr[NH + kk], ov = gr[2 * NH + kk], uv = gr[3 * NH + kk];
                    const float tcv = a.tc[b * NH + kk], cmv = a.cm[b * NH + kk];
                    const float dh = dh_s[e][kk] + a.dhout[b * NH + kk];
                    const float dov = dh * tcv;
                    const float dc = dc_s[e][kk] + dh * ov * (1.f - tcv * tcv);
                    dzi = (dc * uv) * iv * (1.f - iv);
                    dzf = (dc * cmv) * fv * (1.f - fv);
                    dzo = dov * ov * (1.f - ov);
                    dzu = (dc * iv) * (1.f - uv * uv);
                    const float keep = a.mask[a.srow ? a.srow[b] : b] ? 0.f : 1.f;
                    dcm = (dc * fv) * keep;
                    float* dst = a.dzg + b * 4 * NH;
                    dst[kk] = dzi; dst[NH + kk] = dzf; dst[2 * NH + kk] = dzo; dst[3 * NH + kk] = dzu;
                }
                dz_s[e][kk] = dzi; dz_s[e][NH + kk] = dzf; dz_s[e][2 * NH + kk] = dzo; dz_s[e][3 * NH + kk] = dzu;
                dc_s[e][kk] = dcm;
            }
            __syncthreads();
            float acc[E];
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < NH; j4 += 4) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float4 dv = *reinterpret_cast<const float4*>(&dz_s[e][q4 * NH + j4]);
                    acc[e] = fmaf(dv.x, w[j4], acc[e]);
                    acc[e] = fmaf(dv.y, w[j4 + 1], acc[e]);
                    acc[e] = fmaf(dv.z, w[j4 + 2], acc[e]);
                    acc[e] = fmaf(dv.w, w[j4 + 3], acc[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) p_s[e][q4][k] = acc[e];
            __syncthreads();
            for (int q = tid; q < E * NH; q += 4 * NH) {
                const int e = q / NH, kk = q - e * NH, env = g0 + e;
                float keep = 0.f;
                if (env < a.nenv) {
                    const long b = (long)env * T + t;
                    keep = a.mask[a.srow ? a.srow[b] : b] ? 0.f : 1.f;
                }
                dh_s[e][kk] = ((p_s[e][0][kk] + p_s[e][1][kk]) + (p_s[e][2][kk] + p_s[e][3][kk])) * keep;
            }
            __syncthreads();
        }
    }
}

inline bool lstm_nh_ok(int nh) { return nh == 32 || nh == 64 || nh == 128; }

inline hipError_t launch_lstm_fwd(const LstmFwdArgs& a, int nh, int num_cus, hipStream_t st) {
    constexpr int E = 4;
    const int groups = (a.nenv + E - 1) / E;
    const int blocks = std::max(1, std::min(groups, 2 * num_cus));
    if (nh == 128) hipLaunchKernelGGL((lstm_fwd_kernel<128, E>), dim3(blocks), dim3(512), 0, st, a);
    else if (nh == 64) hipLaunchKernelGGL((lstm_fwd_kernel<64, E>), dim3(blocks), dim3(256), 0, st, a);
    else if (nh == 32) hipLaunchKernelGGL((lstm_fwd_kernel<32, E>), dim3(blocks), dim3(128), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
inline hipError_t launch_lstm_bwd(const LstmBwdArgs& a, int nh, int num_cus, hipStream_t st) {
    constexpr int E = 4;
    const int groups = (a.nenv + E - 1) / E;
    const int blocks = std::max(1, std::min(groups, 2 * num_cus));
    if (nh == 128) hipLaunchKernelGGL((lstm_bwd_kernel<128, E>), dim3(blocks), dim3(512), 0, st, a);
    else if (nh == 64) hipLaunchKernelGGL((lstm_bwd_kernel<64, E>), dim3(blocks), dim3(256), 0, st, a);
    else if (nh == 32) hipLaunchKernelGGL((lstm_bwd_kernel<32, E>), dim3(blocks), dim3(128), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace mrl
