// Transposed-accumulator epilogues of the split-bf16 engines (gfx950), and the pre-split "plane tensor" experiment.
//
// (1) What the product uses.  With the MFMA operands swapped (D^T = B A^T) a lane of the 32 x 32 accumulator owns ONE
// output row and the 16 columns {8g + 4h + j : g, j < 4} (h = lane >> 5) of a 32-column block, four consecutive ones per
// accumulator quad.  The epilogues of the hidden layers' forward kernels (h = relu(z + b), common/models.py:15-26 via
// a2c/utils.py:37-63) and of the data gradients (dz = dh * relu'(h), the tf.gradients chain of ppo2/model.py:102-103)
// then store float4s (16 instead of 64 store instructions per thread and tile), assemble the ReLU mask word inside the
// lane (one shuffle per block instead of 16 ballots) and read ONE mask word per 32 channels: c2.fwd 6.3 -> 5.5 ms,
// c2.dgrad 6.2 -> 5.75 ms (profiles/README.md).  Same products, same sums per output as the row-major epilogues.
//
// (2) The experiment (option act_planes bits 1 / 2 / 32, off by default).  The tiled split engines multiply fp32 x fp32 as
// 8 exact bf16 products; splitting the activation operand while it is staged costs ~176 of the ~210 VALU instructions of
// a k step, and the im2col overlap makes conv2 split every element of its input 4 times, conv3 9 times.  Here the kernel
// that PRODUCES an activation also splits each element once in its epilogue and writes a "plane tensor" next to the fp32
// tensor; the consumer's staging is then a plain 16-byte copy (same truncation split, split2_bf16x3: same products).
// Plane tensor of X[rows][C] (C % 32 == 0): three bf16 arrays P[plane][rows][C]; inside each aligned block of 32
// elements along C, element c sits at position perm32(c), which maps a lane's 16 columns to 16 CONSECUTIVE positions
// (2 x 16 bytes per lane and plane, no cross-lane traffic); the consumers' weight planes use the same order of k
// (split_planes_kernel / dgx6_split_planes_kernel, kperm) -- a dot product does not care in which order k runs.
// Measured: the k loops shrink from 204 to 36 VALU instructions per 64 MFMAs and every kernel gets SLOWER (c2.fwd 6.3 ->
// 9.0 ms): a (row, plane, k tile) piece is 64 bytes = half a cache line, the producers write 2.5x the bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hip.h"
#include "wres.hip.h"      // split2_bf16x3

namespace mrl {

__host__ __device__ __forceinline__ int perm32(int c) { return ((c >> 2) & 1) * 16 + (c >> 3) * 4 + (c & 3); }
__host__ __device__ __forceinline__ long kperm32(long k) { return (k & ~31L) | perm32((int)(k & 31)); }

typedef uint32_t pl_u32x4 __attribute__((ext_vector_type(4)));

// Plane-tensor addressing.  Planar (round 2): element e of plane pl at pl * pstride + e -- a (row, plane, 32-wide k tile) piece
// is 64 bytes, half a cache line.  Block-interleaved (round 3): the three planes of each aligned 32-element block sit next
// to each other, [block][plane][32] = 192 contiguous bytes per (row, k tile): element e of plane pl at
// 3 * (e & ~31) + 32 * pl + (e & 31).
constexpr bool kPlanesInterleaved = true;
__host__ __device__ __forceinline__ long plane_off(long e32 /* offset of an aligned 32-block */, int pl, long pstride) {
    return kPlanesInterleaved ? 3 * e32 + 32 * pl : pl * pstride + e32;
}

// ---- epilogue functors of the transposed accumulator layout ---------------------------------------------------------
// o = element offset of the block's first output in the fp32 tensor (a multiple of 32; the same offset in each plane),
// cb = its first column, h = lane >> 5.  A kernel's epilogue runs in two passes over its accumulators: first the
// auxiliary loads of ALL of them (load_aux: bias / activation of the layer below / its mask word -- one memory round trip
// for the whole epilogue instead of one per accumulator), then arithmetic and stores (tr_block_epilogue).
struct TrAux { float4 v[4]; uint32_t w; };

struct TrBiasRelu {          // h = relu(acc + bias[c])  (+ ReLU bit mask of h, + planes of h); bias 16-byte aligned
    float* out; long ld; const float* bias; uint32_t* mask; uint16_t* hp; long pstride;
    __device__ __forceinline__ TrAux load_aux(long, int cb, int h, bool valid) const {
        TrAux x;
        x.w = 0;
        const float* b = bias + (valid ? cb + 4 * h : 4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) x.v[g] = *reinterpret_cast<const float4*>(b + 8 * g);
        return x;
    }
    // sg = +-1: the sign this row's A operand was staged with (x6_dither); folded into the bias add
    __device__ __forceinline__ float4 apply(const TrAux& x, int g, int, float4 a, float sg) const {
        float4 v;
        v.x = act_fwd(__builtin_fmaf(a.x, sg, x.v[g].x), ACT_RELU); v.y = act_fwd(__builtin_fmaf(a.y, sg, x.v[g].y), ACT_RELU);
        v.z = act_fwd(__builtin_fmaf(a.z, sg, x.v[g].z), ACT_RELU); v.w = act_fwd(__builtin_fmaf(a.w, sg, x.v[g].w), ACT_RELU);
        return v;
    }
};
struct TrPartial {           // plain partial sums of a split-K launch (combined by splitk_bias_act_kernel, model.hip)
    float* out; long ld;
    static constexpr uint32_t* mask = nullptr;
    static constexpr uint16_t* hp = nullptr;
    static constexpr long pstride = 0;
    __device__ __forceinline__ TrAux load_aux(long, int, int, bool) const { TrAux x; x.w = 0; return x; }
    __device__ __forceinline__ float4 apply(const TrAux&, int, int, float4 a, float sg) const {
        a.x *= sg; a.y *= sg; a.z *= sg; a.w *= sg;
        return a;
    }
};
struct TrMaskRelu {          // dz = acc * relu'(h) with h the fp32 output of the layer below, or its bit mask, or nothing
    float* out; long ld; const float* h; const uint32_t* hbits; uint16_t* hp; long pstride;
    static constexpr uint32_t* mask = nullptr;
    __device__ __forceinline__ TrAux load_aux(long o, int, int hh, bool valid) const {
        TrAux x;
        x.w = (hbits && valid) ? hbits[o >> 5] : 0u;
        if (!hbits && h) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                x.v[g] = valid ? *reinterpret_cast<const float4*>(h + o + 8 * g + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return x;
    }
    // sg = +-1: the sign this row's A operand was staged with (x6_dither); folded into the mask factor
    __device__ __forceinline__ float4 apply(const TrAux& x, int g, int cin, float4 a, float sg) const {
        if (hbits) {
            a.x *= ((x.w >> cin) & 1u) ? sg : 0.f; a.y *= ((x.w >> (cin + 1)) & 1u) ? sg : 0.f;
            a.z *= ((x.w >> (cin + 2)) & 1u) ? sg : 0.f; a.w *= ((x.w >> (cin + 3)) & 1u) ? sg : 0.f;
        } else if (h) {
            a.x *= x.v[g].x > 0.f ? sg : 0.f; a.y *= x.v[g].y > 0.f ? sg : 0.f;
            a.z *= x.v[g].z > 0.f ? sg : 0.f; a.w *= x.v[g].w > 0.f ? sg : 0.f;
        } else {
            a.x *= sg; a.y *= sg; a.z *= sg; a.w *= sg;
        }
        return a;
    }
};

// ---- whole-line stores out of the transposed layout (round 6) -------------------------------------------------------------
// In the transposed layout lane (i, h) holds, for ITS row i, the four 16-byte chunks q = 2 g + h (g < 4) of a 128-byte group of 32
// columns; store instruction g of the plain epilogue therefore touches 32 rows x 32 bytes: 32 cache lines per instruction, each line
// completed by four instructions.  The CU's vector-memory path works per touched LINE (a timing experiment that only changed the
// addresses -- 8 whole lines per instruction, wrong placement -- took c2.dgrad from 4.77 to 4.27 ms; profiles/README.md, round 6),
// so the four lanes of a quad (rows i0 .. i0+3, same h) exchange their chunks first: a 4 x 4 transpose of 16-byte items over the
// quad's lanes with DPP quad permutes (no LDS).  Afterwards lane (i0 + m, h) holds chunk q = 2 m + h of the rows i0 + s, s < 4, and
// store instruction s writes row i0 + s: per instruction 8 rows x 8 chunks = 8 whole 128-byte lines.  Same values to the same
// addresses.
__device__ __forceinline__ float dpp_quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm:[1,0,3,2]
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm:[2,3,0,1]
}
// x[g] of quad lane m  ->  x[s] = what quad lane s held in x[m]   (per component; odd1 = m & 1, odd2 = m & 2)
__device__ __forceinline__ void quad_transpose4(float& x0, float& x1, float& x2, float& x3, bool odd1, bool odd2) {
    float r;
    r = dpp_quad_xor1(odd1 ? x0 : x1); x0 = odd1 ? r : x0; x1 = odd1 ? x1 : r;
    r = dpp_quad_xor1(odd1 ? x2 : x3); x2 = odd1 ? r : x2; x3 = odd1 ? x3 : r;
    r = dpp_quad_xor2(odd2 ? x0 : x2); x0 = odd2 ? r : x0; x2 = odd2 ? x2 : r;
    r = dpp_quad_xor2(odd2 ? x1 : x3); x1 = odd2 ? r : x1; x3 = odd2 ? x3 : r;
}
__device__ __forceinline__ void quad_transpose4(float4 (&v)[4], int lane) {
    const bool o1 = lane & 1, o2 = lane & 2;
    quad_transpose4(v[0].x, v[1].x, v[2].x, v[3].x, o1, o2);
    quad_transpose4(v[0].y, v[1].y, v[2].y, v[3].y, o1, o2);
    quad_transpose4(v[0].z, v[1].z, v[2].z, v[3].z, o1, o2);
    quad_transpose4(v[0].w, v[1].w, v[2].w, v[3].w, o1, o2);
}

// One 32 x 32 accumulator in the transposed layout: lane (i, h) owns row i and the columns 8g + 4h + j = acc[4g + j] of
// the 32-column block whose first element has offset o (a multiple of 32) in the output tensor.  Writes the fp32 values
// (ef.out), the plane tensor (ef.hp) and the ReLU bit mask (ef.mask: one word per block, bit = column).  All 64 lanes
// must call it (the two halves of a block exchange their mask bits); `valid` gates the memory accesses.
// row_ld > 0 (round 6): the rows i of the block are row_ld elements apart in the output (o = row * row_ld + first column); the fp32
// values then leave as WHOLE 128-byte lines -- the quad's lanes exchange chunks (quad_transpose4) and store instruction s writes
// row i0 + s of every quad, 8 rows x 8 chunks per instruction instead of 32 rows x 2 chunks.  Same values to the same addresses.
template <class EF>
__device__ __forceinline__ void tr_block_epilogue(const EF& ef, const f32x16& acc, const TrAux& aux, long o, int h, bool valid, float sg = 1.f,
                                                  long row_ld = 0) {
    uint32_t bits = 0;
    uint32_t pk[3][8];
    float4 vv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int cin = 8 * g + 4 * h;
        const float4 v = ef.apply(aux, g, cin, make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]), sg);
        vv[g] = v;
        if (ef.mask)
            bits |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << cin;
        if (ef.out && valid && row_ld <= 0) *reinterpret_cast<float4*>(ef.out + o + cin) = v;
        if (ef.hp) {
            split2_bf16x3(v.x, v.y, pk[0][2 * g], pk[1][2 * g], pk[2][2 * g]);
            split2_bf16x3(v.z, v.w, pk[0][2 * g + 1], pk[1][2 * g + 1], pk[2][2 * g + 1]);
        }
    }
    if (ef.out && row_ld > 0) {
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int m = lane & 3;
        // which rows of this quad are inside the output, and where its first row starts (a lane whose own row is outside still stores
        // its chunk of the rows that are inside)
        const unsigned long long bal = __ballot(valid);
        const uint32_t qv = (uint32_t)(bal >> (lane & ~3)) & 0xfu;
        const int olo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)o, 0x00, 0xf, 0xf, true);            // quad_perm:[0,0,0,0]
        const int ohi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((unsigned long long)o >> 32), 0x00, 0xf, 0xf, true);
        const long o0 = (long)(((unsigned long long)(uint32_t)ohi << 32) | (uint32_t)olo);
        quad_transpose4(vv, lane);
        float* d = ef.out + o0 + 8 * m + 4 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if ((qv >> s) & 1u) *reinterpret_cast<float4*>(d + s * row_ld) = vv[s];
    }
    if (ef.hp && valid) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            uint16_t* d = ef.hp + plane_off(o, pl, ef.pstride) + 16 * h;
            *reinterpret_cast<pl_u32x4*>(d) = pl_u32x4{pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]};
            *reinterpret_cast<pl_u32x4*>(d + 8) = pl_u32x4{pk[pl][4], pk[pl][5], pk[pl][6], pk[pl][7]};
        }
    }
    if (ef.mask) {
        const uint32_t other = (uint32_t)__shfl_xor((int)bits, 32);
        if (valid && h == 0) ef.mask[o >> 5] = bits | other;
    }
}

// fp32 tensor -> plane tensor as a separate pass (small tensors whose producer is not one of the MFMA engines: the
// pre-activation gradient of the last hidden layer, written by the heads kernel).  n % 8 == 0, src 16-byte aligned.
__global__ __launch_bounds__(256) void planes_from_f32_kernel(const float* __restrict__ src, long n, uint16_t* __restrict__ hp, long pstride) {
    // a thread converts one half block: 16 consecutive POSITIONS p0 .. p0+15 of a block = columns 8g + 4hh + j
    const long nhalf = n / 16;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < nhalf; e += (long)gridDim.x * 256L) {
        const long blk = e >> 1;
        const int hh = (int)(e & 1);
        uint32_t pk[3][8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(src + blk * 32 + 8 * g + 4 * hh);
            split2_bf16x3(v.x, v.y, pk[0][2 * g], pk[1][2 * g], pk[2][2 * g]);
            split2_bf16x3(v.z, v.w, pk[0][2 * g + 1], pk[1][2 * g + 1], pk[2][2 * g + 1]);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            uint16_t* d = hp + plane_off(blk * 32, pl, pstride) + 16 * hh;
            *reinterpret_cast<pl_u32x4*>(d) = pl_u32x4{pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]};
            *reinterpret_cast<pl_u32x4*>(d + 8) = pl_u32x4{pk[pl][4], pk[pl][5], pk[pl][6], pk[pl][7]};
        }
    }
}
inline hipError_t launch_planes_from_f32(const float* src, long n, uint16_t* hp, long pstride, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const long nhalf = n / 16;
    const int blocks = (int)std::min<long>((nhalf + 255) / 256, 8192);
    hipLaunchKernelGGL(planes_from_f32_kernel, dim3(blocks), dim3(256), 0, stream, src, n, hp, pstride);
    return hipGetLastError();
}

}  // namespace mrl
