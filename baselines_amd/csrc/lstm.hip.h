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
#include <stdlib.h>

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
typedef float lstm_v4f __attribute__((ext_vector_type(4)));

// Index arithmetic is 32-bit on purpose (B * 4 nh elements < 2^32, checked by the launcher): with 64-bit element offsets the
// compiler keeps a 64-bit address pair per (array, environment) alive across the scan -- ~100 VGPRs next to the nh weights,
// which made the nh = 128 instantiation spill 305 registers.  Unsigned 32-bit offsets against the uniform base pointers
// become scalar-base + vector-offset accesses.
template <int NH, int E>
__global__ __launch_bounds__(4 * NH) void lstm_fwd_kernel(LstmFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float h_s[E][NH];
    __shared__ float c_s[E][NH];
    __shared__ float g_s[E][4 * NH];
    const unsigned j = threadIdx.x;
    constexpr unsigned N4 = 4u * NH;
    float w[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) w[k] = a.wh[(unsigned)k * N4 + j];
    const float bj = a.bias[j];
    const unsigned T = (unsigned)a.T, nenv = (unsigned)a.nenv;
    // this thread's (environment, unit) of the element-wise phases: E * NH elements over 4 NH threads
    static_assert(E == 4, "one element-wise item per thread");
    const unsigned ee = j / NH, ek = j - ee * NH;
    for (unsigned g0 = blockIdx.x * E; g0 < nenv; g0 += gridDim.x * E) {
        const unsigned env = g0 + ee;
        const bool live = env < nenv;
        {
            const bool ok = live && a.s0;
            c_s[ee][ek] = ok ? a.s0[env * 2u * NH + ek] : 0.f;
            h_s[ee][ek] = ok ? a.s0[env * 2u * NH + NH + ek] : 0.f;
        }
        __syncthreads();
        for (unsigned t = 0; t < T; ++t) {
            float zxv[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const unsigned en = min(g0 + (unsigned)e, nenv - 1u);
                zxv[e] = a.zx[(en * T + t) * N4 + j];                          // in flight during the mask phase
            }
            const unsigned b = env * T + t;                                    // sample of this thread's element-wise item
            if (live) {
                const float keep = a.mask[a.srow ? (unsigned)a.srow[b] : b] ? 0.f : 1.f;
                const float cv = c_s[ee][ek] * keep, hv = h_s[ee][ek] * keep;
                c_s[ee][ek] = cv;
                h_s[ee][ek] = hv;
                if (a.cm) { a.cm[b * NH + ek] = cv; a.hm[b * NH + ek] = hv; }
            }
            __syncthreads();
            // z[e][j] = sum_k h[e][k] * wh[k][j] for the E = 4 environments at once: v_mfma_f32_4x4x1_16b_f32 is 16 independent
            // 4 x 4 rank-1 updates -- block = 4 consecutive lanes, A row i = environment i (supplied by lane 4*block + i),
            // B column = the lane's own gate column, D[r] = accumulator of environment r in that lane.  One instruction does
            // the four fmaf of a (k, column) pair (same fp32 fma, same k order: bitwise the scalar chain), and the h operand
            // is a per-lane register: a lane reads h[lane & 3][k4 .. k4+3] with ONE 16-byte LDS read per 4 k -- a quarter of
            // the broadcast reads of the fmaf form, which were what a step waited for (128 x E ds_read_b128 per wave).
            lstm_v4f acc4 = {0.f, 0.f, 0.f, 0.f};
            const float* hrow = &h_s[j & 3u][0];
#pragma unroll
            for (int k4 = 0; k4 < NH; k4 += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(hrow + k4);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(hv.x, w[k4], acc4, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(hv.y, w[k4 + 1], acc4, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(hv.z, w[k4 + 2], acc4, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(hv.w, w[k4 + 3], acc4, 0, 0, 0);
            }
            const float acc[E] = {acc4[0], acc4[1], acc4[2], acc4[3]};
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float z = (zxv[e] + acc[e]) + bj;                      // (x@wx + h@wh) + b, the reference's order
                const float gv = j < 3 * NH ? lstm_sigmoid(z) : tanhf(z);
                g_s[e][j] = gv;
                if (a.gates && g0 + e < nenv) a.gates[((g0 + e) * T + t) * N4 + j] = gv;
            }
            __syncthreads();
            {
                const float iv = g_s[ee][ek], fv = g_s[ee][NH + ek], ov = g_s[ee][2 * NH + ek], uv = g_s[ee][3 * NH + ek];
                const float c = fv * c_s[ee][ek] + iv * uv;
                const float tcv = tanhf(c);
                const float h = ov * tcv;
                c_s[ee][ek] = c;
                h_s[ee][ek] = h;
                if (live) {
                    if (a.tc) a.tc[b * NH + ek] = tcv;
                    a.hout[b * NH + ek] = h;
                }
            }
            __syncthreads();
        }
        if (a.s_out && live) {
            a.s_out[env * 2u * NH + ek] = c_s[ee][ek];
            a.s_out[env * 2u * NH + NH + ek] = h_s[ee][ek];
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
    static_assert(E == 4 && NH % 4 == 0, "4x4x1 MFMA blocks: four environments, four lanes of a block in one gate quarter");
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
            // p[e][q4][k] = sum_jj dz[e][q4*NH + jj] * wh[k][q4*NH + jj] on the 4x4x1 MFMA (see lstm_fwd_kernel): the four lanes
            // of a block share q4 (NH % 4 == 0) and supply dz of environments 0..3
            lstm_v4f acc4 = {0.f, 0.f, 0.f, 0.f};
            const float* drow = &dz_s[tid & 3][q4 * NH];
#pragma unroll
            for (int j4 = 0; j4 < NH; j4 += 4) {
                const float4 dv = *reinterpret_cast<const float4*>(drow + j4);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(dv.x, w[j4], acc4, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(dv.y, w[j4 + 1], acc4, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(dv.z, w[j4 + 2], acc4, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(dv.w, w[j4 + 3], acc4, 0, 0, 0);
            }
            const float acc[E] = {acc4[0], acc4[1], acc4[2], acc4[3]};
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

// ---- one environment per workgroup (round 4).  A step of the scans above costs a workgroup ~3.7 us whatever it carries:
// 128 dependent 4x4x1 MFMAs per wave are 2048 matrix-pipe cycles per SIMD, for four environments or for one, and a
// 64-environment minibatch (cnn_lstm, N = 256, 4 minibatches) is 16 workgroups on 256 CUs.  With ONE environment per
// workgroup the product h@wh is 128 plain fmaf per thread (512 VALU cycles per wave, half the matrix-pipe time of the
// 4x4x1 form, which computes four rows whether they are used or not), 64 workgroups run side by side, and the thread that
// owns unit k in the element-wise phases is the same in every phase, so the masked state moves to the next step in
// registers: two barriers per step instead of three.  Same fmaf chains in the same k order, same gate functions: the
// results are bit-identical to lstm_fwd_kernel / lstm_bwd_kernel.  Chosen by the launchers when the scan is long and
// groups of four environments would leave CUs idle.
template <int NH>
__global__ __launch_bounds__(4 * NH) void lstm_fwd1_kernel(LstmFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float h_s[NH];
    __shared__ float g_s[4 * NH];
    const unsigned j = threadIdx.x;
    constexpr unsigned N4 = 4u * NH;
    float w[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) w[k] = a.wh[(unsigned)k * N4 + j];
    const float bj = a.bias[j];
    const unsigned T = (unsigned)a.T, nenv = (unsigned)a.nenv;
    const bool unit = j < (unsigned)NH;                   // this thread also owns unit j of the element-wise phases
    for (unsigned env = blockIdx.x; env < nenv; env += gridDim.x) {
        float c = 0.f, hreg = 0.f;
        if (unit && a.s0) { c = a.s0[env * 2u * NH + j]; hreg = a.s0[env * 2u * NH + NH + j]; }
        if (unit) {                                       // mask of step 0 (a2c/utils.py:89-90)
            const unsigned b = env * T;
            const float keep = a.mask[a.srow ? (unsigned)a.srow[b] : b] ? 0.f : 1.f;
            c *= keep; hreg *= keep;
            h_s[j] = hreg;
            if (a.cm) { a.cm[b * NH + j] = c; a.hm[b * NH + j] = hreg; }
        }
        __syncthreads();
        for (unsigned t = 0; t < T; ++t) {
            const unsigned b = env * T + t;
            const float zxv = a.zx[b * N4 + j];                                   // in flight during the product
            uint8_t mnext = 0;
            if (unit && t + 1 < T) mnext = a.mask[a.srow ? (unsigned)a.srow[b + 1] : b + 1];
            float acc = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < NH; k4 += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(h_s + k4);     // broadcast read
                acc = fmaf(hv.x, w[k4], acc);
                acc = fmaf(hv.y, w[k4 + 1], acc);
                acc = fmaf(hv.z, w[k4 + 2], acc);
                acc = fmaf(hv.w, w[k4 + 3], acc);
            }
            const float z = (zxv + acc) + bj;                                     // (x@wx + h@wh) + b, the reference's order
            const float gv = j < 3 * NH ? lstm_sigmoid(z) : tanhf(z);
            g_s[j] = gv;
            if (a.gates) a.gates[b * N4 + j] = gv;
            __syncthreads();
            if (unit) {
                const float iv = g_s[j], fv = g_s[NH + j], ov = g_s[2 * NH + j], uv = g_s[3 * NH + j];
                c = fv * c + iv * uv;
                const float tcv = tanhf(c);
                hreg = ov * tcv;
                if (a.tc) a.tc[b * NH + j] = tcv;
                a.hout[b * NH + j] = hreg;
                if (t + 1 < T) {                          // the next step's mask, applied here: its masked state is this phase's output
                    const float keep = mnext ? 0.f : 1.f;
                    c *= keep; hreg *= keep;
                    if (a.cm) { a.cm[(b + 1) * NH + j] = c; a.hm[(b + 1) * NH + j] = hreg; }
                }
                h_s[j] = hreg;
            }
            __syncthreads();
        }
        if (a.s_out && unit) {
            a.s_out[env * 2u * NH + j] = c;
            a.s_out[env * 2u * NH + NH + j] = hreg;
        }
    }
}

template <int NH>
__global__ __launch_bounds__(4 * NH) void lstm_bwd1_kernel(LstmBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float dz_s[4 * NH];
    __shared__ float p_s[4][NH];
    const unsigned tid = threadIdx.x;
    const unsigned k = tid % NH, q4 = tid / NH;
    constexpr unsigned N4 = 4u * NH;
    float w[NH];                                  // wh[k][q4*NH + jj]
#pragma unroll
    for (int jj = 0; jj < NH; ++jj) w[jj] = a.wh[k * N4 + q4 * NH + jj];
    const unsigned T = (unsigned)a.T, nenv = (unsigned)a.nenv;
    const bool unit = tid < (unsigned)NH;         // owner of unit tid in the element-wise phase
    for (unsigned env = blockIdx.x; env < nenv; env += gridDim.x) {
        float dhc = 0.f, dcc = 0.f;               // dL/dh and dL/dc carried from step t+1 (this thread's unit)
        for (int ti = (int)T - 1; ti >= 0; --ti) {
            const unsigned b = env * T + (unsigned)ti;
            if (unit) {
                const float* gr = a.gates + b * N4;
                const float iv = gr[tid], fv = gr[NH + tid], ov = gr[2 * NH + tid], uv = gr[3 * NH + tid];
                const float tcv = a.tc[b * NH + tid], cmv = a.cm[b * NH + tid];
                const float dh = dhc + a.dhout[b * NH + tid];
                const float dov = dh * tcv;
                const float dc = dcc + dh * ov * (1.f - tcv * tcv);
                const float dzi = (dc * uv) * iv * (1.f - iv);
                const float dzf = (dc * cmv) * fv * (1.f - fv);
                const float dzo = dov * ov * (1.f - ov);
                const float dzu = (dc * iv) * (1.f - uv * uv);
                float* dst = a.dzg + b * N4;
                dst[tid] = dzi; dst[NH + tid] = dzf; dst[2 * NH + tid] = dzo; dst[3 * NH + tid] = dzu;
                dz_s[tid] = dzi; dz_s[NH + tid] = dzf; dz_s[2 * NH + tid] = dzo; dz_s[3 * NH + tid] = dzu;
                dcc = dc * fv;                    // (* keep below, with dh)
            }
            __syncthreads();
            // p[q4][k] = sum_jj dz[q4*NH + jj] * wh[k][q4*NH + jj]: the lanes of a wave share q4 -> broadcast reads
            float acc = 0.f;
            const float* drow = dz_s + q4 * NH;
#pragma unroll
            for (int j4 = 0; j4 < NH; j4 += 4) {
                const float4 dv = *reinterpret_cast<const float4*>(drow + j4);
                acc = fmaf(dv.x, w[j4], acc);
                acc = fmaf(dv.y, w[j4 + 1], acc);
                acc = fmaf(dv.z, w[j4 + 2], acc);
                acc = fmaf(dv.w, w[j4 + 3], acc);
            }
            p_s[q4][k] = acc;
            __syncthreads();
            if (unit) {
                const float keep = a.mask[a.srow ? (unsigned)a.srow[b] : b] ? 0.f : 1.f;
                dhc = ((p_s[0][tid] + p_s[1][tid]) + (p_s[2][tid] + p_s[3][tid])) * keep;
                dcc = dcc * keep;
            }
            // (p_s is rewritten only behind the next step's first barrier, dz_s only by the unit owners that have just read p_s)
        }
        __syncthreads();
    }
}

// ---- any width: the same scans with wh STREAMED from L2 every step instead of held in registers (nh = 256 is 1 MB of
// weights, more than a CU's whole register file) -- the path of `nlstm` values other than 32 / 64 / 96 / 128 (e.g. the 256 of
// the reference's impala_cnn_lstm, common/models.py:212-214).  Same arithmetic, same order of the fmaf chains.
template <int E>
__global__ __launch_bounds__(1024) void lstm_fwd_generic_kernel(LstmFwdArgs a, int nh) {
    extern __shared__ __attribute__((aligned(16))) float lg_s[];
    float* h_s = lg_s;                       // [E][nh]
    float* c_s = h_s + E * nh;               // [E][nh]
    float* g_s = c_s + E * nh;               // [E][4nh]
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    const unsigned NH = (unsigned)nh, N4 = 4u * NH;
    const unsigned T = (unsigned)a.T, nenv = (unsigned)a.nenv;
    for (unsigned g0 = blockIdx.x * E; g0 < nenv; g0 += gridDim.x * E) {
        for (unsigned q = tid; q < E * NH; q += nt) {
            const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
            const bool ok = env < nenv && a.s0;
            c_s[q] = ok ? a.s0[env * 2u * NH + k] : 0.f;
            h_s[q] = ok ? a.s0[env * 2u * NH + NH + k] : 0.f;
        }
        __syncthreads();
        for (unsigned t = 0; t < T; ++t) {
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
                if (env < nenv) {
                    const unsigned b = env * T + t;
                    const float keep = a.mask[a.srow ? (unsigned)a.srow[b] : b] ? 0.f : 1.f;
                    const float cv = c_s[q] * keep, hv = h_s[q] * keep;
                    c_s[q] = cv;
                    h_s[q] = hv;
                    if (a.cm) { a.cm[b * NH + k] = cv; a.hm[b * NH + k] = hv; }
                }
            }
            __syncthreads();
            for (unsigned j = tid; j < N4; j += nt) {
                float acc[E];
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] = 0.f;
                for (unsigned k = 0; k < NH; ++k) {
                    const float w = a.wh[k * N4 + j];                       // coalesced over j, L2-resident
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] = fmaf(h_s[e * NH + k], w, acc[e]);
                }
                const float bj = a.bias[j];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const unsigned env = min(g0 + (unsigned)e, nenv - 1u);
                    const float z = (a.zx[(env * T + t) * N4 + j] + acc[e]) + bj;
                    const float gv = j < 3 * NH ? lstm_sigmoid(z) : tanhf(z);
                    g_s[e * N4 + j] = gv;
                    if (a.gates && g0 + e < nenv) a.gates[((g0 + e) * T + t) * N4 + j] = gv;
                }
            }
            __syncthreads();
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
                const float* g = g_s + e * N4;
                const float c = g[NH + k] * c_s[q] + g[k] * g[3 * NH + k];
                const float tcv = tanhf(c);
                const float h = g[2 * NH + k] * tcv;
                c_s[q] = c;
                h_s[q] = h;
                if (env < nenv) {
                    const unsigned b = env * T + t;
                    if (a.tc) a.tc[b * NH + k] = tcv;
                    a.hout[b * NH + k] = h;
                }
            }
            __syncthreads();
        }
        if (a.s_out)
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
                if (env < nenv) {
                    a.s_out[env * 2u * NH + k] = c_s[q];
                    a.s_out[env * 2u * NH + NH + k] = h_s[q];
                }
            }
        __syncthreads();
    }
}

template <int E>
__global__ __launch_bounds__(1024) void lstm_bwd_generic_kernel(LstmBwdArgs a, int nh) {
    extern __shared__ __attribute__((aligned(16))) float lg_s[];
    float* dh_s = lg_s;                      // [E][nh]
    float* dc_s = dh_s + E * nh;             // [E][nh]
    float* dz_s = dc_s + E * nh;             // [E][4nh]
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    const unsigned NH = (unsigned)nh, N4 = 4u * NH;
    const unsigned T = (unsigned)a.T, nenv = (unsigned)a.nenv;
    for (unsigned g0 = blockIdx.x * E; g0 < nenv; g0 += gridDim.x * E) {
        for (unsigned q = tid; q < E * NH; q += nt) { dh_s[q] = 0.f; dc_s[q] = 0.f; }
        __syncthreads();
        for (int ti = (int)T - 1; ti >= 0; --ti) {
            const unsigned t = (unsigned)ti;
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, kk = q - e * NH, env = g0 + e;
                float dzi = 0.f, dzf = 0.f, dzo = 0.f, dzu = 0.f, dcm = 0.f;
                if (env < nenv) {
                    const unsigned b = env * T + t;
                    const float* gr = a.gates + b * N4;
                    const float iv = gr[kk], fv = gr[NH + kk], ov = gr[2 * NH + kk], uv = gr[3 * NH + kk];
                    const float tcv = a.tc[b * NH + kk], cmv = a.cm[b * NH + kk];
                    const float dh = dh_s[q] + a.dhout[b * NH + kk];
                    const float dov = dh * tcv;
                    const float dc = dc_s[q] + dh * ov * (1.f - tcv * tcv);
                    dzi = (dc * uv) * iv * (1.f - iv);
                    dzf = (dc * cmv) * fv * (1.f - fv);
                    dzo = dov * ov * (1.f - ov);
                    dzu = (dc * iv) * (1.f - uv * uv);
                    const float keep = a.mask[a.srow ? (unsigned)a.srow[b] : b] ? 0.f : 1.f;
                    dcm = (dc * fv) * keep;
                    float* dst = a.dzg + b * N4;
                    dst[kk] = dzi; dst[NH + kk] = dzf; dst[2 * NH + kk] = dzo; dst[3 * NH + kk] = dzu;
                }
                float* dz = dz_s + e * N4;
                dz[kk] = dzi; dz[NH + kk] = dzf; dz[2 * NH + kk] = dzo; dz[3 * NH + kk] = dzu;
                dc_s[q] = dcm;
            }
            __syncthreads();
            // dh_prev[e][k] = sum over the 4 gate blocks q4 (in the fixed order of the register kernel: ((p0 + p1) + (p2 + p3)))
            // of sum_jj dz[e][q4*nh + jj] * wh[k][q4*nh + jj]
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
                float pq[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float* wrow = a.wh + k * N4 + q4 * NH;
                    const float* dz = dz_s + e * N4 + q4 * NH;
                    float acc = 0.f;
                    for (unsigned jj = 0; jj < NH; ++jj) acc = fmaf(dz[jj], wrow[jj], acc);
                    pq[q4] = acc;
                }
                float keep = 0.f;
                if (env < nenv) {
                    const unsigned b = env * T + t;
                    keep = a.mask[a.srow ? (unsigned)a.srow[b] : b] ? 0.f : 1.f;
                }
                dh_s[q] = ((pq[0] + pq[1]) + (pq[2] + pq[3])) * keep;      // read back by this thread only (next step)
            }
            __syncthreads();
        }
    }
}

// ---- layer-normalised cell --- a2c/utils.py:104-140 `lnlstm` (lstm(layer_norm=True), cnn_lnlstm) ------------------------------
//
//     z = LN(x@wx; gx, bx) + LN(h@wh; gh, bh) + b;   c = f*c + i*u;   h = o * tanh(LN(c; gc, bc))
//     LN(v; g, b) = (v - mean(v)) / sqrt(var(v) + 1e-5) * g + b   over the features of one row (biased variance)
//
// LN(x@wx) is row-parallel and is applied to the GEMM output outside (model.hip); inside the scan a step has two more row
// statistics -- over the 4nh values of h@wh and over the nh values of the new c -- each taken by one wave per environment
// from LDS (two-pass: mean, then mean of squared differences, like tf.nn.moments).  The scan stores the normalised values
// and 1/sqrt(var + e) of both; the backward scan adds the two LN Jacobians to the chain and writes, next to the gate
// gradients dz (which feed dgx/dbx, dgh/dbh, db and the x path outside), the gradient w.r.t. the raw h@wh (for dwh and
// dh_prev) and w.r.t. LN(c) (for dgc/dbc).  Streamed-weight form for every width (the LN variants are not the headline).
constexpr float LNLSTM_EPS = 1e-5f;
struct LnLstmFwdArgs {
    LstmFwdArgs base;       // base.zx already holds LN(x@wx)*gx + bx; base.bias = b
    const float *gh, *bh;   // [4nh]
    const float *gc, *bc;   // [nh]
    float *xhh, *ish;       // [B][4nh] normalised h@wh, [B] 1/sqrt(var+e)      (nullptr on the act side)
    float *xhc, *isc;       // [B][nh]  normalised c,    [B]
};
struct LnLstmBwdArgs {
    LstmBwdArgs base;       // base.tc = tanh(LN(c)), base.dzg = dL/dz out
    const float *gh, *gc;
    const float *xhh, *ish, *xhc, *isc;
    float* dzh;             // [B][4nh] gradient w.r.t. the raw h@wh
    float* dcn;             // [B][nh]  gradient w.r.t. LN(c)*gc + bc
};

// one wave: sum of f(c) over c = lane, lane+64, ... < n, broadcast to every lane
template <class F>
__device__ __forceinline__ float lstm_wave_sum(unsigned n, F f) {
    float s = 0.f;
    for (unsigned c = threadIdx.x & 63u; c < n; c += 64u) s += f(c);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

template <int E>
__global__ __launch_bounds__(1024) void lnlstm_fwd_kernel(LnLstmFwdArgs p, int nh) {
    const LstmFwdArgs& a = p.base;
    extern __shared__ __attribute__((aligned(16))) float lg_s[];
    float* h_s = lg_s;                       // [E][nh]
    float* c_s = h_s + E * nh;               // [E][nh]
    float* g_s = c_s + E * nh;               // [E][4nh]: raw h@wh, then the gates
    float* st_s = g_s + E * 4 * nh;          // [E][2]: mean, 1/sqrt(var + e)
    const unsigned tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6;
    const unsigned NH = (unsigned)nh, N4 = 4u * NH;
    const unsigned T = (unsigned)a.T, nenv = (unsigned)a.nenv;
    for (unsigned g0 = blockIdx.x * E; g0 < nenv; g0 += gridDim.x * E) {
        for (unsigned q = tid; q < E * NH; q += nt) {
            const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
            const bool ok = env < nenv && a.s0;
            c_s[q] = ok ? a.s0[env * 2u * NH + k] : 0.f;
            h_s[q] = ok ? a.s0[env * 2u * NH + NH + k] : 0.f;
        }
        __syncthreads();
        for (unsigned t = 0; t < T; ++t) {
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
                if (env < nenv) {
                    const unsigned b = env * T + t;
                    const float keep = a.mask[a.srow ? (unsigned)a.srow[b] : b] ? 0.f : 1.f;
                    const float cv = c_s[q] * keep, hv = h_s[q] * keep;
                    c_s[q] = cv;
                    h_s[q] = hv;
                    if (a.cm) { a.cm[b * NH + k] = cv; a.hm[b * NH + k] = hv; }
                }
            }
            __syncthreads();
            for (unsigned j = tid; j < N4; j += nt) {
                float acc[E];
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] = 0.f;
                for (unsigned k = 0; k < NH; ++k) {
                    const float w = a.wh[k * N4 + j];
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] = fmaf(h_s[e * NH + k], w, acc[e]);
                }
#pragma unroll
                for (int e = 0; e < E; ++e) g_s[e * N4 + j] = acc[e];
            }
            __syncthreads();
            if (wave < E) {
                const float* row = g_s + wave * N4;
                const float mean = lstm_wave_sum(N4, [&](unsigned c) { return row[c]; }) / (float)N4;
                const float var = lstm_wave_sum(N4, [&](unsigned c) { const float d = row[c] - mean; return d * d; }) / (float)N4;
                if ((tid & 63u) == 0) { st_s[2 * wave] = mean; st_s[2 * wave + 1] = 1.f / sqrtf(var + LNLSTM_EPS); }
            }
            __syncthreads();
            for (unsigned j = tid; j < N4; j += nt) {
                const float ghj = p.gh[j], bhj = p.bh[j], bj = a.bias[j];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool live = g0 + e < nenv;
                    const unsigned b = min(g0 + (unsigned)e, nenv - 1u) * T + t;
                    const float xh = (g_s[e * N4 + j] - st_s[2 * e]) * st_s[2 * e + 1];
                    const float z = (a.zx[b * N4 + j] + (xh * ghj + bhj)) + bj;
                    const float gv = j < 3 * NH ? lstm_sigmoid(z) : tanhf(z);
                    g_s[e * N4 + j] = gv;
                    if (live && a.gates) { a.gates[b * N4 + j] = gv; p.xhh[b * N4 + j] = xh; }
                    if (live && p.ish && j == 0) p.ish[b] = st_s[2 * e + 1];
                }
            }
            __syncthreads();
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, k = q - e * NH;
                const float* g = g_s + e * N4;
                c_s[q] = g[NH + k] * c_s[q] + g[k] * g[3 * NH + k];
            }
            __syncthreads();
            if (wave < E) {
                const float* row = c_s + wave * NH;
                const float mean = lstm_wave_sum(NH, [&](unsigned c) { return row[c]; }) / (float)NH;
                const float var = lstm_wave_sum(NH, [&](unsigned c) { const float d = row[c] - mean; return d * d; }) / (float)NH;
                if ((tid & 63u) == 0) { st_s[2 * wave] = mean; st_s[2 * wave + 1] = 1.f / sqrtf(var + LNLSTM_EPS); }
            }
            __syncthreads();
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
                const float xc = (c_s[q] - st_s[2 * e]) * st_s[2 * e + 1];
                const float tcv = tanhf(xc * p.gc[k] + p.bc[k]);
                const float h = g_s[e * N4 + 2 * NH + k] * tcv;
                h_s[q] = h;
                if (env < nenv) {
                    const unsigned b = env * T + t;
                    if (a.tc) { a.tc[b * NH + k] = tcv; p.xhc[b * NH + k] = xc; if (k == 0) p.isc[b] = st_s[2 * e + 1]; }
                    a.hout[b * NH + k] = h;
                }
            }
            __syncthreads();
        }
        if (a.s_out)
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
                if (env < nenv) {
                    a.s_out[env * 2u * NH + k] = c_s[q];
                    a.s_out[env * 2u * NH + NH + k] = h_s[q];
                }
            }
        __syncthreads();
    }
}

template <int E>
__global__ __launch_bounds__(1024) void lnlstm_bwd_kernel(LnLstmBwdArgs p, int nh) {
    const LstmBwdArgs& a = p.base;
    extern __shared__ __attribute__((aligned(16))) float lg_s[];
    float* dh_s = lg_s;                      // [E][nh]
    float* dc_s = dh_s + E * nh;             // [E][nh]
    float* dz_s = dc_s + E * nh;             // [E][4nh]: dL/dz, then dL/d(raw h@wh)
    float* r_s = dz_s + E * 4 * nh;          // [E][nh]: dL/d(normalised c)
    float* st_s = r_s + E * nh;              // [E][2]: the two row means of an LN Jacobian
    const unsigned tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6;
    const unsigned NH = (unsigned)nh, N4 = 4u * NH;
    const unsigned T = (unsigned)a.T, nenv = (unsigned)a.nenv;
    for (unsigned g0 = blockIdx.x * E; g0 < nenv; g0 += gridDim.x * E) {
        for (unsigned q = tid; q < E * NH; q += nt) { dh_s[q] = 0.f; dc_s[q] = 0.f; }
        __syncthreads();
        for (int ti = (int)T - 1; ti >= 0; --ti) {
            const unsigned t = (unsigned)ti;
            // h = o * tanh(cn), cn = xhat_c * gc + bc:  dL/dcn, dL/dxhat_c
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, kk = q - e * NH, env = g0 + e;
                float dx = 0.f;
                if (env < nenv) {
                    const unsigned b = env * T + t;
                    const float ov = a.gates[b * N4 + 2 * NH + kk], tcv = a.tc[b * NH + kk];
                    const float dh = dh_s[q] + a.dhout[b * NH + kk];
                    const float dcn = dh * ov * (1.f - tcv * tcv);
                    p.dcn[b * NH + kk] = dcn;
                    dx = dcn * p.gc[kk];
                    dh_s[q] = dh;                                            // total dL/dh_t, used again below
                }
                r_s[q] = dx;
            }
            __syncthreads();
            if (wave < E) {
                const unsigned env = g0 + wave;
                float m1 = 0.f, m2 = 0.f;
                if (env < nenv) {
                    const float* row = r_s + wave * NH;
                    const float* xh = p.xhc + (env * T + t) * NH;
                    m1 = lstm_wave_sum(NH, [&](unsigned c) { return row[c]; }) / (float)NH;
                    m2 = lstm_wave_sum(NH, [&](unsigned c) { return row[c] * xh[c]; }) / (float)NH;
                }
                if ((tid & 63u) == 0) { st_s[2 * wave] = m1; st_s[2 * wave + 1] = m2; }
            }
            __syncthreads();
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, kk = q - e * NH, env = g0 + e;
                float dzi = 0.f, dzf = 0.f, dzo = 0.f, dzu = 0.f, dcm = 0.f;
                if (env < nenv) {
                    const unsigned b = env * T + t;
                    const float* gr = a.gates + b * N4;
                    const float iv = gr[kk], fv = gr[NH + kk], ov = gr[2 * NH + kk], uv = gr[3 * NH + kk];
                    const float tcv = a.tc[b * NH + kk], cmv = a.cm[b * NH + kk];
                    const float dh = dh_s[q];
                    const float dov = dh * tcv;
                    const float dc = dc_s[q] + p.isc[b] * ((r_s[q] - st_s[2 * e]) - p.xhc[b * NH + kk] * st_s[2 * e + 1]);
                    dzi = (dc * uv) * iv * (1.f - iv);
                    dzf = (dc * cmv) * fv * (1.f - fv);
                    dzo = dov * ov * (1.f - ov);
                    dzu = (dc * iv) * (1.f - uv * uv);
                    const float keep = a.mask[a.srow ? (unsigned)a.srow[b] : b] ? 0.f : 1.f;
                    dcm = (dc * fv) * keep;
                    float* dst = a.dzg + b * N4;
                    dst[kk] = dzi; dst[NH + kk] = dzf; dst[2 * NH + kk] = dzo; dst[3 * NH + kk] = dzu;
                }
                float* dz = dz_s + e * N4;
                dz[kk] = dzi; dz[NH + kk] = dzf; dz[2 * NH + kk] = dzo; dz[3 * NH + kk] = dzu;
                dc_s[q] = dcm;
            }
            __syncthreads();
            // the h path: z contains LN(h@wh)*gh + bh -> Jacobian of that LN over the 4nh features
            if (wave < E) {
                const unsigned env = g0 + wave;
                float m1 = 0.f, m2 = 0.f;
                if (env < nenv) {
                    const float* row = dz_s + wave * N4;
                    const float* xh = p.xhh + (env * T + t) * N4;
                    m1 = lstm_wave_sum(N4, [&](unsigned c) { return row[c] * p.gh[c]; }) / (float)N4;
                    m2 = lstm_wave_sum(N4, [&](unsigned c) { return row[c] * p.gh[c] * xh[c]; }) / (float)N4;
                }
                if ((tid & 63u) == 0) { st_s[2 * wave] = m1; st_s[2 * wave + 1] = m2; }
            }
            __syncthreads();
            for (unsigned j = tid; j < N4; j += nt) {
                const float ghj = p.gh[j];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    float v = 0.f;
                    if (g0 + e < nenv) {
                        const unsigned b = (g0 + e) * T + t;
                        v = p.ish[b] * ((dz_s[e * N4 + j] * ghj - st_s[2 * e]) - p.xhh[b * N4 + j] * st_s[2 * e + 1]);
                        p.dzh[b * N4 + j] = v;
                    }
                    dz_s[e * N4 + j] = v;
                }
            }
            __syncthreads();
            for (unsigned q = tid; q < E * NH; q += nt) {
                const unsigned e = q / NH, k = q - e * NH, env = g0 + e;
                float pq[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float* wrow = a.wh + k * N4 + q4 * NH;
                    const float* dz = dz_s + e * N4 + q4 * NH;
                    float acc = 0.f;
                    for (unsigned jj = 0; jj < NH; ++jj) acc = fmaf(dz[jj], wrow[jj], acc);
                    pq[q4] = acc;
                }
                float keep = 0.f;
                if (env < nenv) {
                    const unsigned b = env * T + t;
                    keep = a.mask[a.srow ? (unsigned)a.srow[b] : b] ? 0.f : 1.f;
                }
                dh_s[q] = ((pq[0] + pq[1]) + (pq[2] + pq[3])) * keep;
            }
            __syncthreads();
        }
    }
}
inline bool lnlstm_nh_ok(int nh) { return nh >= 1 && nh <= 512; }
inline hipError_t launch_lnlstm_fwd(const LnLstmFwdArgs& p, int nh, int num_cus, hipStream_t st) {
    constexpr int E = 4;
    if (!lnlstm_nh_ok(nh)) return hipErrorInvalidValue;
    const int groups = (p.base.nenv + E - 1) / E;
    const int blocks = std::max(1, std::min(groups, 2 * num_cus));
    const int threads = std::max(64 * E, std::min(1024, (4 * nh + 63) / 64 * 64));       // at least one wave per environment
    hipLaunchKernelGGL((lnlstm_fwd_kernel<E>), dim3(blocks), dim3(threads), ((size_t)E * nh * 6 + 2 * E) * sizeof(float), st, p, nh);
    return hipGetLastError();
}
inline hipError_t launch_lnlstm_bwd(const LnLstmBwdArgs& p, int nh, int num_cus, hipStream_t st) {
    constexpr int E = 4;
    if (!lnlstm_nh_ok(nh)) return hipErrorInvalidValue;
    const int groups = (p.base.nenv + E - 1) / E;
    const int blocks = std::max(1, std::min(groups, 2 * num_cus));
    const int threads = std::max(64 * E, std::min(1024, (4 * nh + 63) / 64 * 64));
    hipLaunchKernelGGL((lnlstm_bwd_kernel<E>), dim3(blocks), dim3(threads), ((size_t)E * nh * 7 + 2 * E) * sizeof(float), st, p, nh);
    return hipGetLastError();
}

// widths the register-resident kernels are compiled for; any other positive width takes the streamed form
inline bool lstm_nh_fast(int nh) { return nh == 32 || nh == 64 || nh == 96 || nh == 128; }
inline bool lstm_nh_ok(int nh) { return nh >= 1 && nh <= 1024; }

inline int& lstm_e1() { static int v = getenv("MRL_LSTM_E1") ? atoi(getenv("MRL_LSTM_E1")) : 1; return v; }     // mrl_set_option "lstm_e1"
// one environment per workgroup: long scans whose groups of four environments would leave CUs idle
inline bool lstm_use_e1(int nenv, int T, int nh, int num_cus) {
    // the one-environment kernels index with unsigned 32-bit element offsets (row * 4nh): longer rollouts take the 4-env kernels
    return lstm_e1() && (nh == 128 || nh == 64) && T >= 8 && (nenv + 3) / 4 < num_cus && (long)nenv * T * 4 * nh < (1L << 32);
}
inline hipError_t launch_lstm_fwd(const LstmFwdArgs& a, int nh, int num_cus, hipStream_t st) {
    constexpr int E = 4;
    const int groups = (a.nenv + E - 1) / E;
    const int blocks = std::max(1, std::min(groups, 2 * num_cus));
    if (lstm_use_e1(a.nenv, a.T, nh, num_cus)) {
        const int b1 = std::max(1, std::min(a.nenv, 2 * num_cus));
        if (nh == 128) hipLaunchKernelGGL((lstm_fwd1_kernel<128>), dim3(b1), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((lstm_fwd1_kernel<64>), dim3(b1), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (!lstm_nh_fast(nh)) {
        if (!lstm_nh_ok(nh)) return hipErrorInvalidValue;
        const int threads = std::min(1024, (4 * nh + 63) / 64 * 64);
        hipLaunchKernelGGL((lstm_fwd_generic_kernel<E>), dim3(blocks), dim3(threads), (size_t)E * nh * 6 * sizeof(float), st, a, nh);
        return hipGetLastError();
    }
    if (nh == 128) hipLaunchKernelGGL((lstm_fwd_kernel<128, E>), dim3(blocks), dim3(512), 0, st, a);
    else if (nh == 96) hipLaunchKernelGGL((lstm_fwd_kernel<96, E>), dim3(blocks), dim3(384), 0, st, a);
    else if (nh == 64) hipLaunchKernelGGL((lstm_fwd_kernel<64, E>), dim3(blocks), dim3(256), 0, st, a);
    else if (nh == 32) hipLaunchKernelGGL((lstm_fwd_kernel<32, E>), dim3(blocks), dim3(128), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
inline hipError_t launch_lstm_bwd(const LstmBwdArgs& a, int nh, int num_cus, hipStream_t st) {
    constexpr int E = 4;
    const int groups = (a.nenv + E - 1) / E;
    const int blocks = std::max(1, std::min(groups, 2 * num_cus));
    if (lstm_use_e1(a.nenv, a.T, nh, num_cus)) {
        const int b1 = std::max(1, std::min(a.nenv, 2 * num_cus));
        if (nh == 128) hipLaunchKernelGGL((lstm_bwd1_kernel<128>), dim3(b1), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((lstm_bwd1_kernel<64>), dim3(b1), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (!lstm_nh_fast(nh)) {
        if (!lstm_nh_ok(nh)) return hipErrorInvalidValue;
        const int threads = std::min(1024, (E * nh + 63) / 64 * 64);
        hipLaunchKernelGGL((lstm_bwd_generic_kernel<E>), dim3(blocks), dim3(threads), (size_t)E * nh * 6 * sizeof(float), st, a, nh);
        return hipGetLastError();
    }
    if (nh == 128) hipLaunchKernelGGL((lstm_bwd_kernel<128, E>), dim3(blocks), dim3(512), 0, st, a);
    else if (nh == 96) hipLaunchKernelGGL((lstm_bwd_kernel<96, E>), dim3(blocks), dim3(384), 0, st, a);
    else if (nh == 64) hipLaunchKernelGGL((lstm_bwd_kernel<64, E>), dim3(blocks), dim3(256), 0, st, a);
    else if (nh == 32) hipLaunchKernelGGL((lstm_bwd_kernel<32, E>), dim3(blocks), dim3(128), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace mrl
