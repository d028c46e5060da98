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
#include <type_traits>

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
#ifndef C1_TR_STORE_STRIDE
#define C1_TR_STORE_STRIDE 6
#endif

struct C1TrRelu {            // h = relu(acc) (bias folded into the accumulator), fp32 + bit mask + planes
    float* out; uint32_t* mask; uint16_t* hp; long pstride;
    __device__ __forceinline__ float4 apply(const TrAux&, int, int, float4 a, float) const {
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

// ---- third generation (round 4): the second-generation kernel walks its phases in lock-step -- all 8 waves multiply, then
// all of them walk their epilogue (16 dword stores + ballots per tile), then all convert and stage the next image, then the
// barrier: the matrix pipe idles for ~45 % of an image's time (6.1 k MFMA cycles on the busiest SIMD of ~11 k per image at
// 3.02 ms per 131072 images).  Here the three phases are ONE software pipeline per wave: while the MFMAs of image b run,
// the wave issues -- between them, in program order, a few instructions per MFMA -- (a) the stores / mask ballots of ITS
// tiles of image b-G (kept ReLU'd in 32 registers), (b) the u8 -> bf16 conversion and LDS writes of ITS chunks of image
// b+G (into the other LDS buffer), (c) the global loads of image b+2G.  Nothing but the barrier is left between two MFMA
// streams.  Tiles: 13 per image (12.5 real); waves 0-4 own two, waves 5-7 one each, so every SIMD (waves w, w+4) carries
// 3-4 tiles.  Same products, same order of accumulation per output as c1fwd2_kernel: bit-identical results.
template <bool MASK, int DBG = 0, bool TR = false>
__global__ __launch_bounds__(512) void c1fwd3_kernel(const uint8_t* __restrict__ obs, const int32_t* __restrict__ srow,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ out, uint32_t* __restrict__ mask, int B) {
    constexpr int NT = 512;
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
    float bvr[16];                                        // TR: the accumulator register indexes the channel (8g + 4h + j = register 4g + j)
#pragma unroll
    for (int r = 0; r < 16; ++r) bvr[r] = TR ? bias[2 * (r & ~3) + 4 * h + (r & 3)] : 0.f;
    const int G = gridDim.x;
    c1_u32x4 st[NLD];
    // chunk q of this thread (clamped: the 28 threads past the image's end re-read its last chunk and drop it when staging)
    const int cidx[NLD] = {min(tid, CHUNKS - 1), min(NT + tid, CHUNKS - 1), min(2 * NT + tid, CHUNKS - 1), min(3 * NT + tid, CHUNKS - 1)};
    auto img_ptr = [&](int b) {
        b = min(b, B - 1);                                // past the end: a valid image, staged and never multiplied
        const long row = srow ? (long)srow[b] : (long)b;
        return obs + row * C1_IMG;
    };
    auto stage_half = [&](uint8_t* dst, int q, int half) {       // 8 bytes of chunk q -> 8 bf16 (16 bytes of LDS)
        uint32_t e0, e1, e2, e3;
        u8x4_to_bf16(st[q][2 * half], e0, e1);
        u8x4_to_bf16(st[q][2 * half + 1], e2, e3);
        if (q < NLD - 1 || 3 * NT + tid < CHUNKS)
            *reinterpret_cast<c1_u32x4*>(dst + (long)cidx[q] * 32 + 16 * half) = c1_u32x4{e0, e1, e2, e3};
    };
    int b = blockIdx.x;
    {
        const uint8_t* g = img_ptr(b);
#pragma unroll
        for (int q = 0; q < NLD; ++q) st[q] = *reinterpret_cast<const c1_u32x4*>(g + (long)cidx[q] * 16);
    }
    __syncthreads();                                      // planes built
#pragma unroll
    for (int q = 0; q < NLD; ++q) { stage_half(img, q, 0); stage_half(img, q, 1); }
    {
        const uint8_t* g = img_ptr(b + G);
#pragma unroll
        for (int q = 0; q < NLD; ++q) st[q] = *reinterpret_cast<const c1_u32x4*>(g + (long)cidx[q] * 16);
    }
    __syncthreads();
    const uint8_t* wrow = reinterpret_cast<const uint8_t*>(wp + (long)i * C1_KP + 8 * h);
    // tiles of this wave.  Row-major accumulators (TR = false): waves 0-4 -> (2w, 2w+1), waves 5, 6, 7 -> 10, 11, 12 (the last one:
    // 16 pixels computed as 32).  TR, round 6: an image's phase lasts as long as its busiest SIMD, and SIMD 0 (waves 0 and 4)
    // carried FOUR tiles where the others carry three -- leaving one tile out (timing experiment) took the kernel from 2.76 to 2.29 ms.
    // Now the twelve whole tiles go three to a SIMD (waves 0-3 -> (2w, 2w+1), waves 4-7 -> 8 .. 11) and the 16 pixels 384 .. 399
    // are two 16-pixel x 16-channel blocks on v_mfma_f32_16x16x32_bf16 (24 instructions each, no padding rows), one on wave 4, one
    // on wave 5: 3.25 / 3.25 / 3 / 3 tile times per SIMD instead of 4 / 3 / 3 / 3.
    const int t0 = TR ? (wave < 4 ? 2 * wave : 4 + wave) : (wave < 5 ? 2 * wave : 5 + wave);
    const bool two = TR ? wave < 4 : wave < 5;
    const bool mini = TR && (wave == 4 || wave == 5);
    const int chalf = wave & 1;                            // mini block: channels 16 chalf .. 16 chalf + 15 (wave 4: 0, wave 5: 1)
    const int l16 = lane & 15, kg4 = lane >> 4;           // 16x16x32 operand roles: row / column l16, k group kg4 (8 consecutive k)
    const int mini_aoff = (((384 + l16) / C1_OW * C1_S * C1_W + ((384 + l16) % C1_OW) * C1_S) * C1_C + 8 * kg4) * 2;
    const uint8_t* const mini_w = reinterpret_cast<const uint8_t*>(wp + (long)(16 * chalf + l16) * C1_KP + 8 * kg4);
    typedef float c1_f32x4 __attribute__((ext_vector_type(4)));
    c1_f32x4 pend16 = {0.f, 0.f, 0.f, 0.f};               // ReLU'd mini block of the previous image
    float bias16[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias16[r] = bias[16 * chalf + 4 * kg4 + r];
    int aoff[2];                                          // byte offset of this lane's patch inside an LDS image, per tile
    int pixl[2];                                          // TR: output pixel of this lane, per tile
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        // the last tile has 16 pixels: its lanes 16 .. 31 compute pixels 384 .. 399 AGAIN (TR: they then store the same values to
        // the same addresses as lanes 0 .. 15 -- no predicates in the epilogue)
        int m = (t0 + u) * 32 + i;
        if (m >= C1_PIX) m = TR ? m - 16 : C1_PIX - 1;
        m = min(m, C1_PIX - 1);
        pixl[u] = m;
        const int oy = m / C1_OW, ox = m - oy * C1_OW;
        aoff[u] = ((oy * C1_S * C1_W + ox * C1_S) * C1_C + 8 * h) * 2;
    }
    const int nr0 = (t0 == TILES - 1) ? 8 : 16;           // valid accumulator registers of the first tile (the last tile: pixels 384 .. 399)
    f32x16 pend[2];                                       // ReLU'd outputs of this wave's tiles of the previous image
    long ppix0 = 0;

    // one image: MFMAs with (EPI) the previous image's epilogue, the next image's staging and the loads of the one after that
    // issued between them.  TWO: this wave owns two tiles.
    // the previous image's mini block: lane (l16, kg4) holds pixel 384 + l16, channels 16 chalf + 4 kg4 .. + 3
    auto mini_flush = [&]() {
        *reinterpret_cast<c1_f32x4*>(out + (ppix0 + 384 + l16) * C1_NF + 16 * chalf + 4 * kg4) = pend16;
        if constexpr (MASK) {
            uint32_t bits = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) bits |= min(__float_as_uint(pend16[r]), 1u) << (4 * kg4 + r);
            bits |= (uint32_t)__shfl_xor((int)bits, 16);
            bits |= (uint32_t)__shfl_xor((int)bits, 32);
            if (lane < 16) reinterpret_cast<uint16_t*>(mask)[2 * (ppix0 + 384 + l16) + chalf] = (uint16_t)bits;      // its half of the pixel's word
        }
    };
    auto phase = [&](auto two_c, auto epi_c, auto mini_c, const uint8_t* cur, uint8_t* nxt, int bnext2) {
        constexpr bool TWO = decltype(two_c)::value, EPI = decltype(epi_c)::value, MINI = decltype(mini_c)::value;
        constexpr int NU = TWO ? 2 : 1;
        const uint8_t* arow[2] = {cur + aoff[0], cur + aoff[1]};
        f32x16 acc[2];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = TR ? bvr[r] : bv;  // bias: column (= lane & 31) constant of the C layout (TR: row)
        c1_u32x4 fa[2][2], fb[2][3];
        auto lds_a = [&](int q, int u, c1_u32x4 (&a)[2]) {
            a[u] = *reinterpret_cast<const c1_u32x4*>(arow[u] + ((q >> 1) * (C1_W * C1_C) + 16 * (q & 1)) * 2);
        };
        auto lds_b = [&](int q, int pl, c1_u32x4 (&bq)[3]) {
            bq[pl] = *reinterpret_cast<const c1_u32x4*>(wrow + (q * 16 + pl * 32 * C1_KP) * 2);
        };
#pragma unroll
        for (int u = 0; u < NU; ++u) lds_a(0, u, fa[0]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) lds_b(0, pl, fb[0]);
        float* ob[2];
        int mw[2] = {0, 0};
        if constexpr (EPI) {
#pragma unroll
            for (int u = 0; u < NU; ++u)
                ob[u] = TR ? out + ((ppix0 + pixl[u] - (i & 3)) * C1_NF + 8 * (i & 3) + 4 * h) : out + ((ppix0 + (t0 + u) * 32 + 4 * h) * C1_NF + i);
        }
        // TR epilogue piece e (0 .. 8 NU - 1) of the previous image: tile e >> 3, channel group g = (e & 7) >> 1 (channels
        // 8g + 4h .. +3); even pieces: its 16-byte store; odd pieces: its mask bits (bit = channel; v >= 0 after the ReLU)
        // The four 16-byte stores of a tile complete 32 whole 128-byte lines between them: issued close together (every SS-th
        // gap) they merge on the way to memory; spread evenly over the phase the same stores cost 0.3 ms more per launch.
        // round 6: WHOLE-LINE stores.  Lane (i, h) holds the chunks 2g + h of ITS pixel; piece j of the exchange (epi_tr_xpose) transposes
        // component j of the four chunks over the lanes of a quad (planes.hip.h, quad_transpose4), after which the lane holds chunk
        // 2 (i & 3) + h of the quad's pixels i0 + s in registers 4s .. 4s+3, and store s writes pixel i0 + s: 8 pixels x 128 bytes per
        // instruction instead of 32 pixels x 32 bytes (the CU's vector-memory path works per touched line).  The mask bits are taken
        // BEFORE the exchange (they belong to the lane's own pixel).
        auto epi_tr_xpose = [&](int u, int j) {
            if constexpr (EPI && TR)
            {
                float x0 = pend[u][j], x1 = pend[u][4 + j], x2 = pend[u][8 + j], x3 = pend[u][12 + j];
                quad_transpose4(x0, x1, x2, x3, (i & 1) != 0, (i & 2) != 0);
                pend[u][j] = x0; pend[u][4 + j] = x1; pend[u][8 + j] = x2; pend[u][12 + j] = x3;
            }
        };
        auto epi_tr_store = [&](int u, int sq) {
            if constexpr (EPI && TR) {
                if (!(DBG & 1))
                    *reinterpret_cast<float4*>(ob[u] + sq * C1_NF) = make_float4(pend[u][4 * sq], pend[u][4 * sq + 1], pend[u][4 * sq + 2], pend[u][4 * sq + 3]);
            }
        };
        auto epi_tr_mask = [&](int u, int g) {
            if constexpr (EPI && TR && MASK && !(DBG & 16)) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    mw[u] |= (int)(min(__float_as_uint(pend[u][4 * g + j]), 1u) << (8 * g + j));       // (shifted by 4h once, below)
            }
        };
        const uint8_t* gnext = img_ptr(bnext2);
        // epilogue row e (0 .. 16 NU - 1) of the previous image: tile e >> 4, accumulator register e & 15
        auto epi_row = [&](int e) {
            if constexpr (EPI) {
                const int u = e >> 4, r = e & 15, rr = (r & 3) + 8 * (r >> 2);
                if (u == 0 && r >= nr0) return;           // (wave-uniform) rows past the image's last pixel
                const float v = pend[u][r];
                if (!(DBG & 1)) ob[u][rr * C1_NF] = v;
                if constexpr (MASK && !(DBG & 16)) {
                    const unsigned long long bal = __ballot(v > 0.f);
                    const uint32_t blo = (uint32_t)bal, bhi = (uint32_t)(bal >> 32);
                    // (the s_nop covers the VALU-writes-SGPR -> v_writelane wait states, which the compiler does not insert
                    // around inline assembly)
                    asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
                                 : "+v"(mw[u]) : "s"(blo), "s"(bhi), "n"(rr), "n"(rr + 4));
                }
            }
        };
#pragma unroll
        for (int q = 0; q < 2 * C1_RF; ++q) {
            const int cq = q & 1;
            int slot = 0;                                  // MFMA gap inside this k block
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (!(DBG & 2)) {
                        if constexpr (TR) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[cq][pl]), __builtin_bit_cast(bf16x8, fa[cq][u]), acc[u], 0, 0, 0);
                        else acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cq][u]), __builtin_bit_cast(bf16x8, fb[cq][pl]), acc[u], 0, 0, 0);
                    }
                    else acc[u][0] += __uint_as_float(fa[cq][u][0] ^ fb[cq][pl][0]);       // timing experiment: fragments consumed, no matrix work
                    // ---- the gap behind this MFMA
                    const int gap = q * 3 * NU + slot;     // 0 .. 48 NU - 1
                    if (q + 1 < 2 * C1_RF) {
                        // fragment reads of the next k block, in the order its MFMAs need them (plane 2 first): the fragment
                        // read last is needed ~2 gaps into the next block, every other one has 4-6 gaps of cover
                        if constexpr (TWO) {
                            if (slot == 0) lds_b(q + 1, 2, fb[cq ^ 1]);
                            if (slot == 1) lds_a(q + 1, 0, fa[cq ^ 1]);
                            if (slot == 2) lds_a(q + 1, 1, fa[cq ^ 1]);
                            if (slot == 3) lds_b(q + 1, 1, fb[cq ^ 1]);
                            if (slot == 4) lds_b(q + 1, 0, fb[cq ^ 1]);
                        } else {
                            if (slot == 0) { lds_b(q + 1, 2, fb[cq ^ 1]); lds_a(q + 1, 0, fa[cq ^ 1]); }
                            if (slot == 1) lds_b(q + 1, 1, fb[cq ^ 1]);
                            if (slot == 2) lds_b(q + 1, 0, fb[cq ^ 1]);
                        }
                    }
                    if constexpr (TR) {
                        constexpr int SS = C1_TR_STORE_STRIDE;
                        const int gu = gap % 48, u = gap / 48;                        // 48 gaps per tile
                        // gaps 0 .. 6: mask bits; 8 .. 20: the exchange, one component per four gaps; from 24: the four stores
                        if (gu < 8 && gu % 2 == 0) epi_tr_mask(u, gu / 2);
                        if (gu >= 8 && gu < 24 && gu % 4 == 0) epi_tr_xpose(u, (gu - 8) / 4);
                        if (gu >= 24 && (gu - 24) % SS == 0 && (gu - 24) / SS < 4) epi_tr_store(u, (gu - 24) / SS);
                    }
                    else if (gap % 3 == 0) epi_row(gap / 3);                          // 32 (16) rows over 96 (48) gaps
                    if (gap % 3 == 1 && gap / 3 < 2 * NLD && !(DBG & 8) && !((DBG & 32) && TWO)) stage_half(nxt, gap / 6, (gap / 3) & 1);   // gaps 1, 4, .. 22: 8 half chunks
                    if (gap % 3 == 1 && gap / 3 >= 2 * NLD && gap / 3 < 3 * NLD && !(DBG & 4) && !((DBG & 32) && TWO))
                        st[gap / 3 - 2 * NLD] = *reinterpret_cast<const c1_u32x4*>(gnext + (long)cidx[gap / 3 - 2 * NLD] * 16);
                    __builtin_amdgcn_sched_barrier(0);
                    ++slot;
                }
        }
        if constexpr (EPI && MASK && TR) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int mine = mw[u] << (4 * h);
                const int other = __shfl_xor(mine, 32);    // the other 16 channels of this lane's pixel
                if (h == 0 && !(DBG & 1)) mask[ppix0 + pixl[u]] = (uint32_t)(mine | other);
            }
        } else if constexpr (EPI && MASK) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int t = t0 + u;
                if (lane < (t == TILES - 1 ? 16 : 32)) mask[ppix0 + t * 32 + lane] = (uint32_t)mw[u];
            }
        }
        if constexpr (MINI) {
            if constexpr (EPI) mini_flush();
            c1_f32x4 a4 = {bias16[0], bias16[1], bias16[2], bias16[3]};
#pragma unroll
            for (int ky = 0; ky < C1_RF; ++ky) {             // one patch row = 32 k per instruction
                const c1_u32x4 af = *reinterpret_cast<const c1_u32x4*>(cur + mini_aoff + ky * (C1_W * C1_C) * 2);
#pragma unroll
                for (int pl = 2; pl >= 0; --pl) {
                    const c1_u32x4 wf = *reinterpret_cast<const c1_u32x4*>(mini_w + (ky * 32 + pl * 32 * C1_KP) * 2);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, af), a4, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) pend16[r] = fmaxf(a4[r], 0.f);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) pend[u][r] = fmaxf(acc[u][r], 0.f);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // Control flow: the first image (no epilogue yet) and the second one are peeled, the loop runs from the third.  The image
    // loads of phase k are consumed in phase k+1 behind ~10 younger stores; the compiler's s_waitcnt insertion merges the
    // states of all paths into a loop header, so a loop entered straight from the prologue (loads just issued, nothing
    // behind them) gets vmcnt(0..3) at the top of EVERY phase -- which also waits for the previous phase's last stores
    // (0.6 ms of 2.8 per 131072 images).  Entered from a peeled copy of its own body, the counts are exact (vmcnt(10)).
    int par = 0;
    bool first = true;
#define MRL_C1_STEP(TWOC, EPIC, MINIC)                                                                            \
    {                                                                                                            \
        phase(TWOC, EPIC, MINIC, img + par * C1_IMG16, img + (par ^ 1) * C1_IMG16, b + 2 * G);                    \
        ppix0 = (long)b * C1_PIX;                                                                                \
        first = false;                                                                                           \
        b += G;                                                                                                  \
        par ^= 1;                                                                                                \
        __syncthreads(); /* next image staged; this image's patch reads are done */                              \
    }
    if (two) {
        if (b < B) MRL_C1_STEP(T_{}, F_{}, F_{})
        if (b < B) {
            MRL_C1_STEP(T_{}, T_{}, F_{})
            while (b < B) MRL_C1_STEP(T_{}, T_{}, F_{})
        }
    } else if (mini) {
        if constexpr (TR) {
            if (b < B) MRL_C1_STEP(F_{}, F_{}, T_{})
            if (b < B) {
                MRL_C1_STEP(F_{}, T_{}, T_{})
                while (b < B) MRL_C1_STEP(F_{}, T_{}, T_{})
            }
        }
    } else {
        if (b < B) MRL_C1_STEP(F_{}, F_{}, F_{})
        if (b < B) {
            MRL_C1_STEP(F_{}, T_{}, F_{})
            while (b < B) MRL_C1_STEP(F_{}, T_{}, F_{})
        }
    }
#undef MRL_C1_STEP
    // the last image's epilogue
    if (!first && TR && mini) mini_flush();
    if (!first && TR) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            float* ob = out + ((ppix0 + pixl[u] - (i & 3)) * C1_NF + 8 * (i & 3) + 4 * h);
            uint32_t bits = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) bits |= min(__float_as_uint(pend[u][4 * g + j]), 1u) << (8 * g + 4 * h + j);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x0 = pend[u][j], x1 = pend[u][4 + j], x2 = pend[u][8 + j], x3 = pend[u][12 + j];
                quad_transpose4(x0, x1, x2, x3, (i & 1) != 0, (i & 2) != 0);
                pend[u][j] = x0; pend[u][4 + j] = x1; pend[u][8 + j] = x2; pend[u][12 + j] = x3;
            }
#pragma unroll
            for (int sq = 0; sq < 4; ++sq)
                *reinterpret_cast<float4*>(ob + sq * C1_NF) = make_float4(pend[u][4 * sq], pend[u][4 * sq + 1], pend[u][4 * sq + 2], pend[u][4 * sq + 3]);
            if constexpr (MASK) {
                const uint32_t other = (uint32_t)__shfl_xor((int)bits, 32);
                if (h == 0) mask[ppix0 + pixl[u]] = bits | other;
            }
        }
    } else if (!first) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const int t = t0 + u;
            const int nr = (t == TILES - 1) ? 8 : 16;
            float* ob = out + ((ppix0 + t * 32 + 4 * h) * C1_NF + i);
            int mw = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r < nr) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    const float v = pend[u][r];
                    ob[rr * C1_NF] = v;
                    if constexpr (MASK) {
                        const unsigned long long bal = __ballot(v > 0.f);
                        const uint32_t blo = (uint32_t)bal, bhi = (uint32_t)(bal >> 32);
                        asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
                                     : "+v"(mw) : "s"(blo), "s"(bhi), "n"(rr), "n"(rr + 4));
                    }
                }
            }
            if (MASK && lane < (nr == 16 ? 32 : 16)) mask[ppix0 + t * 32 + lane] = (uint32_t)mw;
        }
    }
}

inline hipError_t launch_c1fwd3(const void* obs, const int32_t* srow, const float* w, const float* bias, float* out, uint32_t* mask,
                                int B, int num_cus, hipStream_t stream, int dbg = 0, bool tr = false) {
    auto kern = mask ? c1fwd3_kernel<true> : c1fwd3_kernel<false>;
    if (tr) kern = mask ? c1fwd3_kernel<true, 0, true> : c1fwd3_kernel<false, 0, true>;
#ifdef MRL_C1_EXPERIMENTS
    if (tr) switch (dbg) {
        case 1: kern = c1fwd3_kernel<true, 1, true>; break;   case 2: kern = c1fwd3_kernel<true, 2, true>; break;
        case 4: kern = c1fwd3_kernel<true, 4, true>; break;   case 8: kern = c1fwd3_kernel<true, 8, true>; break;
        case 16: kern = c1fwd3_kernel<true, 16, true>; break; case 3: kern = c1fwd3_kernel<true, 3, true>; break;
        case 31: kern = c1fwd3_kernel<true, 31, true>; break; case 32: kern = c1fwd3_kernel<true, 32, true>; break;
        default: break;
    } else
    switch (dbg) {        // timing experiments: 1 no stores, 2 no MFMAs, 4 no global loads, 8 no staging, 16 no mask ballots
        case 1: kern = c1fwd3_kernel<true, 1>; break;   case 2: kern = c1fwd3_kernel<true, 2>; break;
        case 4: kern = c1fwd3_kernel<true, 4>; break;   case 8: kern = c1fwd3_kernel<true, 8>; break;
        case 16: kern = c1fwd3_kernel<true, 16>; break; case 17: kern = c1fwd3_kernel<true, 17>; break;
        case 29: kern = c1fwd3_kernel<true, 29>; break; case 31: kern = c1fwd3_kernel<true, 31>; break;
        case 12: kern = c1fwd3_kernel<true, 12>; break; case 3: kern = c1fwd3_kernel<true, 3>; break;
        default: break;
    }
#endif
    { hipError_t e0 = raise_lds_limit((const void*)kern); if (e0 != hipSuccess) return e0; }      // once per (device, kernel)
    const size_t lds = (size_t)3 * 32 * C1_KP * 2 + (size_t)2 * C1_IMG16;          // 163584
    const int grid = std::max(1, std::min(B, num_cus));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, static_cast<const uint8_t*>(obs), srow, w, bias, out, mask, B);
    return hipGetLastError();
}

// hp != nullptr: transposed-accumulator kernel, also writes the plane tensor of the output (pstride elements per plane)
inline hipError_t launch_c1fwd2(const void* obs, const int32_t* srow, const float* w, const float* bias, float* out, uint32_t* mask,
                                int B, int num_cus, hipStream_t stream, uint16_t* hp = nullptr, long pstride = 0, bool tr_plain = false) {
    const bool tr = hp || tr_plain;
    auto kern = tr ? c1fwd2_kernel<0, true> : c1fwd2_kernel<0, false>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
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
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    const int ngroups = (B + G - 1) / G;
    const int grid = std::max(1, std::min(ngroups, num_cus));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), c1fwd_lds_bytes(G), stream, static_cast<const uint8_t*>(obs), srow, w,
                       bias, out, mask, B, dbg);
    return hipGetLastError();
}

}  // namespace mrl
