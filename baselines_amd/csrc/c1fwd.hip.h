// First conv layer of NatureCNN, forward, image-resident (gfx950).  `h = relu(conv(X / 255., 'c1', nf=32, rf=8, stride=4))`
// (common/models.py:19-21 via a2c/utils.py:37-56) for uint8 84x84x4 observations.
//
// The weights-resident engine (wres.hip.h) gathers every A fragment straight from global memory: 16 bytes per lane and
// patch row, each input byte fetched by up to 4 overlapping patches -- 13.4 GB of per-lane gathers for 3.7 GB of pixels,
// texture-addresser bound at 0.21 of the bf16x3 pipe.  Here a persistent workgroup per CU keeps the filter planes AND a
// group of G whole images (28 KB each, uint8) in one of TWO LDS buffers: images arrive once, coalesced, through registers
// (the next group is fetched and written to the other buffer while the current one is multiplied: one barrier per group),
// patches are 8-byte LDS reads, and the output pixels of the group (G * 400) are walked in 32-row MFMA tiles.
// Arithmetic is that of the bf16x3 path: uint8 pixels are exact in bf16, filter/255 is split into three exact bf16
// planes, every product is exact in the fp32 accumulator (v_mfma_f32_32x32x16_bf16).
//   k = (ky, kx, c) in HWIO order; MFMA k block (16 k) = half a patch row = 16 consecutive image bytes;
//   lane (i, h) of a wave = output pixel i of its tile, bytes 8h..8h+7 of the half row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wres.hip.h"
#include "planes.hip.h"

namespace mrl {

constexpr int C1_H = 84, C1_W = 84, C1_C = 4, C1_RF = 8, C1_S = 4, C1_OH = 20, C1_OW = 20, C1_NF = 32;
constexpr int C1_K = C1_RF * C1_RF * C1_C;              // 256
constexpr int C1_KP = C1_K + 8;                         // padded plane row (bf16)
constexpr int C1_IMG = C1_H * C1_W * C1_C;              // 28224 bytes
constexpr int C1_PIX = C1_OH * C1_OW;                   // 400 output pixels per image

template <int G, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void c1fwd_lds_kernel(const uint8_t* __restrict__ obs, const int32_t* __restrict__ srow,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, uint32_t* __restrict__ mask, int B, int dbg) {
    constexpr int NT = WAVES * 64;
    constexpr int CHUNKS = G * C1_IMG / 16;               // 16-byte chunks of a group
    constexpr int NLD = (CHUNKS + NT - 1) / NT;
    constexpr int TILES = (G * C1_PIX + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) uint16_t c1s[];
    uint16_t* wp = c1s;                                   // [3][32][KP] bf16 planes of filter / 255
    uint8_t* img = reinterpret_cast<uint8_t*>(c1s + 3 * 32 * C1_KP);      // [2 buffers][G][C1_IMG]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // ---- filter planes (models.py:19 scale folded into the filter), once per workgroup
    for (int e = tid; e < 3 * 32 * C1_KP; e += NT) wp[e] = 0;
    __syncthreads();
    for (int e = tid; e < C1_K * C1_NF; e += NT) {
        const int k = e / C1_NF, n = e - k * C1_NF;
        const float v = w[e] / 255.f;
        const uint32_t h0 = bf16_rn_bits(v);
        const float r1 = v - __uint_as_float(h0 << 16);
        const uint32_t h1 = bf16_rn_bits(r1);
        const float r2 = r1 - __uint_as_float(h1 << 16);
        const uint32_t h2 = bf16_rn_bits(r2);
        wp[(0 * 32 + n) * C1_KP + k] = (uint16_t)h0;
        wp[(1 * 32 + n) * C1_KP + k] = (uint16_t)h1;
        wp[(2 * 32 + n) * C1_KP + k] = (uint16_t)h2;
    }
    const float bv = bias[i];
    const int ngroups = (B + G - 1) / G;
    uint4 st[NLD];
    auto fetch = [&](int grp) {                           // group -> registers (coalesced 16-byte chunks)
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int c = q * NT + tid;
            const int g = min(c / (C1_IMG / 16), G - 1), off = c - g * (C1_IMG / 16);
            const int b = min(grp * G + g, B - 1);
            const long row = srow ? (long)srow[b] : (long)b;
            st[q] = *reinterpret_cast<const uint4*>(obs + row * C1_IMG + (long)min(off, C1_IMG / 16 - 1) * 16);
        }
    };
    auto stage = [&](uint8_t* dst) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int c = q * NT + tid;
            if (c < CHUNKS) *reinterpret_cast<uint4*>(dst + (long)c * 16) = st[q];
        }
    };
    // two LDS image buffers: group n+1 is fetched into registers and written to the other buffer while group n is
    // multiplied; ONE barrier per group
    int grp = blockIdx.x;
    if (grp < ngroups && !(dbg & 4)) { fetch(grp); }
    __syncthreads();                                      // planes built
    if (grp < ngroups && !(dbg & 4)) stage(img);
    __syncthreads();
    const uint16_t* wrow = wp + (long)i * C1_KP + 8 * h;
    const int mr = (lane & 3) + 4 * (lane >> 3), mh = (lane >> 2) & 1;
    int par = 0;
    for (; grp < ngroups; grp += gridDim.x, par ^= 1) {
        const uint8_t* cur_img = img + par * (G * C1_IMG);
        const bool more = grp + (int)gridDim.x < ngroups;
        if (more && !(dbg & 4)) fetch(grp + gridDim.x);   // next group in flight during the MFMA phase
        const long pix0 = (long)grp * (G * C1_PIX);       // first output pixel of the group in the [B*400] pixel index space
        const long pixN = (long)B * C1_PIX;
        for (int t = wave; t < TILES; t += WAVES) {
            const int p = t * 32 + i;                     // output pixel of the group handled by this lane's row
            const int pc = min(p, G * C1_PIX - 1);
            const int g = pc / C1_PIX, m = pc - g * C1_PIX;
            const int oy = m / C1_OW, ox = m - oy * C1_OW;
            const uint8_t* arow = cur_img + g * C1_IMG + (oy * C1_S * C1_W + ox * C1_S) * C1_C + 8 * h;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // 16 k blocks (ky, half row), software-pipelined by hand: the LDS reads of block q+1 are issued before the three
            // MFMAs of block q (a fully unrolled loop would otherwise be serialised into read -> wait -> MFMA groups)
            uint2 raw[2];
            bf16x8 bfr[2][3];
            auto lds_block = [&](int q, uint2& r, bf16x8 (&bf)[3]) {
                const int ky = q >> 1, blk = q & 1;
                r = *reinterpret_cast<const uint2*>(arow + ky * (C1_W * C1_C) + 16 * blk);
                const uint16_t* wb = wrow + q * 16;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const bf16x8*>(wb + (long)pl * 32 * C1_KP);
            };
            lds_block(0, raw[0], bfr[0]);
#pragma unroll
            for (int q = 0; q < 2 * C1_RF; ++q) {
                const int cur = q & 1;
                if (q + 1 < 2 * C1_RF && !(dbg & 8)) lds_block(q + 1, raw[cur ^ 1], bfr[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
                U32x4 a;
                u8x4_to_bf16(raw[cur].x, a.x, a.y);
                u8x4_to_bf16(raw[cur].y, a.z, a.w);
                const bf16x8 av = __builtin_bit_cast(bf16x8, a);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    if (!(dbg & 2)) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bfr[cur][pl], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: bias + ReLU, 32 channels of a pixel = 128 contiguous bytes; bit mask: one word per pixel.  The images of
            // a group are consecutive samples, so pixel pr of the group is pixel pix0 + pr of the output: no division.
            // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            uint32_t mword = 0;
            const long prow0 = pix0 + t * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                const long gp = prow0 + rr;
                const bool ok = t * 32 + 4 * h + rr < G * C1_PIX && gp < pixN;
                const float v = fmaxf(acc[r] + bv, 0.f);
                if (ok && (!(dbg & 1) || v == 12345.678f)) out[gp * C1_NF + i] = v;
                if (mask) {
                    const unsigned long long bal = __ballot(ok && v > 0.f);
                    if (lane < 32 && mr == r) mword = (uint32_t)(mh ? (bal >> 32) : bal);
                }
            }
            if (mask && lane < 32) {
                const long gp = pix0 + t * 32 + lane;
                if (t * 32 + lane < G * C1_PIX && gp < pixN) mask[gp] = mword;
            }
        }
        if (more && !(dbg & 4)) stage(img + (par ^ 1) * (G * C1_IMG));
        __syncthreads();                                  // next group staged; this group's patch reads are done
    }
}

// ---- second generation: the image is converted to bf16 ONCE while it is staged (every VALU instruction costs the SIMD
// 4 cycles of matrix-pipe issue: scripts/valu_ubench.hip; the kernel above converts every byte ~4 times inside the MFMA
// loop), a wave owns TWO 32-pixel tiles so that a filter fragment is read once per 6 MFMAs, the bias is the initial
// accumulator value, output rows are addressed by instruction immediates and the ReLU mask words are collected with
// v_writelane from the wave ballots.  8 waves, two LDS image buffers (2 x 56448 B bf16) + filter planes (50688 B).
constexpr int C1_IMG16 = C1_IMG * 2;                     // bytes of a bf16 image
typedef uint32_t c1_u32x4 __attribute__((ext_vector_type(4)));

struct C1TrRelu {            // h = relu(acc) (bias folded into the accumulator), fp32 + bit mask + planes
    float* out; uint32_t* mask; uint16_t* hp; long pstride;
    __device__ __forceinline__ float4 apply(const TrAux&, int, int, float4 a) const {
        return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
    }
};
// TR (planes.hip.h): MFMA operands swapped, a lane owns one pixel and 16 of its 32 channels: the epilogue also writes the
// plane tensor of h (the pre-split A operand of conv2's forward) and stores 16 bytes per lane and instruction.
template <int DBG = 0, bool TR = false>
__global__ __launch_bounds__(512) void c1fwd2_kernel(const uint8_t* __restrict__ obs, const int32_t* __restrict__ srow,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ out, uint32_t* __restrict__ mask, int B,
                                                     uint16_t* __restrict__ hp, long pstride) {
    constexpr int NT = 512, WAVES = 8;
    constexpr int CHUNKS = C1_IMG / 16;                   // 1764 16-byte chunks of an image
    constexpr int NLD = (CHUNKS + NT - 1) / NT;           // 4
    constexpr int TILES = (C1_PIX + 31) / 32;             // 13 (the last one: 16 pixels)
    extern __shared__ __attribute__((aligned(16))) uint16_t c1s[];
    uint16_t* wp = c1s;                                   // [3][32][KP] bf16 planes of filter / 255
    uint8_t* img = reinterpret_cast<uint8_t*>(c1s + 3 * 32 * C1_KP);      // [2 buffers][C1_IMG] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    for (int e = tid; e < 3 * 32 * C1_KP; e += NT) wp[e] = 0;
    __syncthreads();
    for (int e = tid; e < C1_K * C1_NF; e += NT) {
        const int k = e / C1_NF, n = e - k * C1_NF;
        const float v = w[e] / 255.f;
        const uint32_t h0 = bf16_rn_bits(v);
        const float r1 = v - __uint_as_float(h0 << 16);
        const uint32_t h1 = bf16_rn_bits(r1);
        const float r2 = r1 - __uint_as_float(h1 << 16);
        const uint32_t h2 = bf16_rn_bits(r2);
        wp[(0 * 32 + n) * C1_KP + k] = (uint16_t)h0;
        wp[(1 * 32 + n) * C1_KP + k] = (uint16_t)h1;
        wp[(2 * 32 + n) * C1_KP + k] = (uint16_t)h2;
    }
    const float bv = bias[i];
    float bvr[16];                                        // TR: the accumulator register indexes the channel
#pragma unroll
    for (int r = 0; r < 16; ++r) bvr[r] = TR ? bias[(r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
    c1_u32x4 st[NLD];
    auto fetch = [&](int b) {
        const long row = srow ? (long)srow[b] : (long)b;
        const uint8_t* g = obs + row * C1_IMG;
#pragma unroll
        for (int q = 0; q < NLD; ++q) st[q] = *reinterpret_cast<const c1_u32x4*>(g + (long)min(q * NT + tid, CHUNKS - 1) * 16);
    };
    auto stage = [&](uint8_t* dst) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int c = q * NT + tid;
            if (c < CHUNKS) {
                uint32_t e[8];
                u8x4_to_bf16(st[q][0], e[0], e[1]);
                u8x4_to_bf16(st[q][1], e[2], e[3]);
                u8x4_to_bf16(st[q][2], e[4], e[5]);
                u8x4_to_bf16(st[q][3], e[6], e[7]);
                *reinterpret_cast<c1_u32x4*>(dst + (long)c * 32) = c1_u32x4{e[0], e[1], e[2], e[3]};
                *reinterpret_cast<c1_u32x4*>(dst + (long)c * 32 + 16) = c1_u32x4{e[4], e[5], e[6], e[7]};
            }
        }
    };
    int b = blockIdx.x;
    if (b < B) fetch(b);
    __syncthreads();                                      // planes built
    if (b < B) stage(img);
    __syncthreads();
    const uint8_t* wrow = reinterpret_cast<const uint8_t*>(wp + (long)i * C1_KP + 8 * h);
    int par = 0;
    for (; b < B; b += gridDim.x, par ^= 1) {
        const uint8_t* cur = img + par * C1_IMG16;
        const bool more = b + (int)gridDim.x < B;
        if (more) fetch(b + gridDim.x);                   // next image in flight during the MFMA phase
        __builtin_amdgcn_sched_barrier(0);
        const int t0 = 2 * wave;
        if (t0 < TILES) {
            const bool two = t0 + 1 < TILES;
            const uint8_t* arow[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int m = min((t0 + u) * 32 + i, C1_PIX - 1);
                const int oy = m / C1_OW, ox = m - oy * C1_OW;
                arow[u] = cur + ((oy * C1_S * C1_W + ox * C1_S) * C1_C + 8 * h) * 2;
            }
            f32x16 acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = TR ? bvr[r] : bv;          // bias: column (= lane & 31) constant of the C layout
            c1_u32x4 fa[2][2], fb[2][3];
            auto lds_block = [&](int q, c1_u32x4 (&a)[2], c1_u32x4 (&bq)[3]) {
                const int off = ((q >> 1) * (C1_W * C1_C) + 16 * (q & 1)) * 2;
                a[0] = *reinterpret_cast<const c1_u32x4*>(arow[0] + off);
                a[1] = *reinterpret_cast<const c1_u32x4*>(arow[1] + off);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bq[pl] = *reinterpret_cast<const c1_u32x4*>(wrow + (q * 16 + pl * 32 * C1_KP) * 2);
            };
            lds_block(0, fa[0], fb[0]);
#pragma unroll
            for (int q = 0; q < 2 * C1_RF; ++q) {
                const int cq = q & 1;
                if (q + 1 < 2 * C1_RF) lds_block(q + 1, fa[cq ^ 1], fb[cq ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pl = 2; pl >= 0; --pl) {
                    if (DBG & 2) continue;
                    if constexpr (TR) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[cq][pl]), __builtin_bit_cast(bf16x8, fa[cq][0]), acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[cq][pl]), __builtin_bit_cast(bf16x8, fa[cq][1]), acc[1], 0, 0, 0);
                    } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cq][0]), __builtin_bit_cast(bf16x8, fb[cq][pl]), acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cq][1]), __builtin_bit_cast(bf16x8, fb[cq][pl]), acc[1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: ReLU, 32 channels of a pixel = 128 contiguous bytes, rows at immediate offsets; mask: one word per pixel =
            // one half of a wave ballot, written into lane (pixel of the tile) by v_writelane.
            // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            const long pix0 = (long)b * C1_PIX;
            if constexpr (TR) {
                // lane (i, h): pixel t*32 + i, channels 8g + 4h + j; the bias is already in the accumulator
                C1TrRelu ef{out, mask, hp, pstride};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && !two) break;
                    const int pix = (t0 + u) * 32 + i;
                    const bool valid = pix < C1_PIX;
                    tr_block_epilogue(ef, acc[u], TrAux{}, valid ? (pix0 + pix) * C1_NF : 0L, h, valid);
                }
            } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                const int t = t0 + u;
                const int nr = (t == TILES - 1) ? 8 : 16;             // last tile: pixels 384 .. 399 = rows 0 .. 15
                float* ob = out + ((pix0 + t * 32 + 4 * h) * C1_NF + i);
                int mw = 0;
                if (mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (r < nr) {
                            const int rr = (r & 3) + 8 * (r >> 2);
                            const float v = fmaxf(acc[u][r], 0.f);
                            if (!(DBG & 1)) ob[rr * C1_NF] = v;
                            const unsigned long long bal = __ballot(v > 0.f);
                            const uint32_t blo = (uint32_t)bal, bhi = (uint32_t)(bal >> 32);
                            // (the s_nop covers the VALU-writes-SGPR -> v_writelane wait states, which the compiler does not
                            // insert around inline assembly)
                            asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
                                         : "+v"(mw) : "s"(blo), "s"(bhi), "n"(rr), "n"(rr + 4));
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (r < nr && !(DBG & 1)) ob[((r & 3) + 8 * (r >> 2)) * C1_NF] = fmaxf(acc[u][r], 0.f);
                }
                if (mask && lane < (nr == 16 ? 32 : 16)) mask[pix0 + t * 32 + lane] = (uint32_t)mw;
            }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) stage(img + (par ^ 1) * C1_IMG16);
        __syncthreads();                                  // next image staged; this image's patch reads are done
    }
}

// hp != nullptr: transposed-accumulator kernel, also writes the plane tensor of the output (pstride elements per plane)
inline hipError_t launch_c1fwd2(const void* obs, const int32_t* srow, const float* w, const float* bias, float* out, uint32_t* mask,
                                int B, int num_cus, hipStream_t stream, uint16_t* hp = nullptr, long pstride = 0, bool tr_plain = false) {
    const bool tr = hp || tr_plain;
    auto kern = tr ? c1fwd2_kernel<0, true> : c1fwd2_kernel<0, false>;
    static bool raised[2] = {false, false};
    if (!raised[tr ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised[tr ? 1 : 0] = true;
    }
    const size_t lds = (size_t)3 * 32 * C1_KP * 2 + (size_t)2 * C1_IMG16;          // 163584
    const int grid = std::max(1, std::min(B, num_cus));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, static_cast<const uint8_t*>(obs), srow, w, bias, out, mask, B, hp, pstride);
    return hipGetLastError();
}

inline size_t c1fwd_lds_bytes(int G) { return (size_t)3 * 32 * C1_KP * 2 + (size_t)2 * G * C1_IMG; }

inline hipError_t launch_c1fwd_lds(const void* obs, const int32_t* srow, const float* w, const float* bias, float* out,
                                   uint32_t* mask, int B, int num_cus, hipStream_t stream, int dbg = 0) {
    constexpr int G = 1, WAVES = 16;
    auto kern = c1fwd_lds_kernel<G, WAVES>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised = true;
    }
    const int ngroups = (B + G - 1) / G;
    const int grid = std::max(1, std::min(ngroups, num_cus));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), c1fwd_lds_bytes(G), stream, static_cast<const uint8_t*>(obs), srow, w,
                       bias, out, mask, B, dbg);
    return hipGetLastError();
}

}  // namespace mrl
