// Weight gradient of the first conv layer of NatureCNN (uint8 84x84x4 observations, 32 filters 8x8 stride 4:
// common/models.py:19-21 via a2c/utils.py:37-56; tf.gradients in ppo2/model.py:100-109), image-resident, gfx950:
//     dW[(ky,kx,c)][n] = 1/255 * sum_{b, oy, ox} X[b][4oy+ky][4ox+kx][c] * dz[b][oy][ox][n],   db[n] = sum dz
// on the bf16 matrix pipe with exact products (uint8 pixels are exact in bf16, dz is split exactly into three bf16
// planes; wres.hip.h "bf16 x 3").  The contraction index is the output PIXEL, and v_mfma_f32_32x32x16_bf16 wants 8
// consecutive contraction elements per lane, so BOTH operands are transposed while they are staged:
//   * image: T[y][x & 3][c][x >> 2] (bf16, converted once per byte in the staging pass) -- 8 consecutive output pixels
//     of one patch byte (kx, c) are 8 consecutive elements: aligned dword reads + v_alignbyte for kx >= 4 instead of 8
//     single-byte gathers and 12 conversion instructions per fragment;
//   * dz: threads load 16-byte pieces of 8 consecutive pixels, split, pack PIXEL pairs and write 16-byte runs of
//     dzT[plane][n][pixel] (the transposed-write trick of wgradx8.hip.h), octets XOR-swizzled against bank conflicts;
//   * a wave owns 2 patch rows ky (2 accumulators) and half of the pixels (13 / 12 blocks of 16): a dz fragment is read
//     once per 6 MFMAs (the predecessor read it once per 3 and gathered bytes: LDS-bound);
//   * the next image's pixels and dz are in flight (registers) during the MFMA phase.
// One persistent workgroup per CU, one partial slab per workgroup (summed by reduce_slabs, model.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "c1fwd.hip.h"
#include "gemmx6.hip.h"

namespace mrl {

constexpr int CW_XI = 24;                                // elements per (y, x&3, c) run: 21 used
constexpr int CW_TROW = 16 * CW_XI * 2 + 40;             // bytes per image row of T (bf16): 202 dwords = 10 mod 64, so the
                                                         // staging writes of a wave (13 rows x 5 runs of 8 bytes) spread over the banks
constexpr int CW_TBYTES = C1_H * CW_TROW;                // 67872
constexpr int CW_PP = 432;                               // bf16 per dzT row (400 pixels; 216 dwords = 24 mod 64)
constexpr int CW_PLANE = C1_NF * CW_PP;
constexpr size_t CW_LDS = (size_t)CW_TBYTES + (size_t)3 * CW_PLANE * 2;       // 150816
constexpr int CW_NT = 512;

template <int DBG = 0>       // timing experiments (-DMRL_X6_EXPERIMENTS, option c1_dbg = 32 + bits): 1 no MFMA phase, 2 no staging pass, 4 no loads, 8 no MFMAs, 16 no image staging, 32 no dz staging
__global__ __launch_bounds__(CW_NT) void c1wgrad_kernel(const uint8_t* __restrict__ obs, const int32_t* __restrict__ srow,
                                                        const float* __restrict__ dz, int B, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) uint8_t cws[];
    uint8_t* T = cws;                                     // bf16 [y][x & 3][c][x >> 2]
    uint16_t* dzt = reinterpret_cast<uint16_t*>(cws + CW_TBYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    constexpr int NA = 2, NKG = C1_RF / NA, NPG = 8 / NKG;       // accumulators (patch rows) per wave; 25 blocks over NPG pixel groups
    constexpr int NB0 = (25 + NPG - 1) / NPG, NB1 = 25 / NPG;  // blocks of group 0 / of the others: 13 + 12 (7 + 6 + 6 + 6 for NA = 4)
    static_assert(NB0 + (NPG - 1) * NB1 == 25, "block split");
    const int kg = wave % NKG, pg = wave / NKG;          // patch rows NA*kg .. NA*kg + NA-1, pixel group

    // ---- staging roles
    const bool img_full = tid < 420, img_small = tid >= 420 && tid < 504;
    const int iy = img_full ? tid / 5 : (img_small ? tid - 420 : 0), iq = img_full ? tid % 5 : 5;
    const int img_off = (iy * C1_W + 16 * iq) * C1_C;    // 16 pixels (64 bytes) of row iy; iq = 5: the last 4 pixels
    const bool dz_on = tid < 400;
    const int d_o = dz_on ? tid >> 3 : 0, d_nc = tid & 7;
    u32x4v vi[4];
    float4 vd[8];
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch_img = [&](long row) {
        const uint8_t* gi = obs + row * C1_IMG + img_off;
        vi[0] = *reinterpret_cast<const u32x4v*>(gi);
        if (img_full) {
#pragma unroll
            for (int j = 1; j < 4; ++j) vi[j] = *reinterpret_cast<const u32x4v*>(gi + 16 * j);
        }
    };
    auto fetch_dz = [&](int b) {
        const float* gd = dz + ((long)b * C1_PIX + 8 * d_o) * C1_NF + 4 * d_nc;
#pragma unroll
        for (int r = 0; r < 8; ++r) vd[r] = *reinterpret_cast<const float4*>(gd + r * C1_NF);
    };
    auto image_row = [&](int b) { return b < B ? (srow ? (long)srow[b] : (long)b) : 0L; };
    // pixels are converted to bf16 HERE, once per image byte (the MFMA phase would convert every byte ~4 times, and
    // every VALU instruction costs the SIMD 4 cycles of matrix-pipe issue)
    auto stage_img = [&]() {
        if (DBG & 16) {
        } else if (img_full) {
            uint8_t* d = T + iy * CW_TROW + 8 * iq;
#pragma unroll
            for (int p = 0; p < 4; ++p)                   // pixel x = 16 iq + 4 j + p: phase p, run index 4 iq + j
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t f0 = __float_as_uint((float)((vi[0][p] >> (8 * c)) & 0xff));
                    const uint32_t f1 = __float_as_uint((float)((vi[1][p] >> (8 * c)) & 0xff));
                    const uint32_t f2 = __float_as_uint((float)((vi[2][p] >> (8 * c)) & 0xff));
                    const uint32_t f3 = __float_as_uint((float)((vi[3][p] >> (8 * c)) & 0xff));
                    *reinterpret_cast<uint2*>(d + (p * 4 + c) * (CW_XI * 2)) =
                        make_uint2(__builtin_amdgcn_perm(f1, f0, 0x07060302u), __builtin_amdgcn_perm(f3, f2, 0x07060302u));
                }
        } else if (img_small) {
            uint8_t* d = T + iy * CW_TROW + 40;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *reinterpret_cast<uint16_t*>(d + (p * 4 + c) * (CW_XI * 2)) =
                        (uint16_t)(__float_as_uint((float)((vi[0][p] >> (8 * c)) & 0xff)) >> 16);
        }
    };
    auto stage_dz = [&]() {
        if (dz_on && !(DBG & 32)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t p[3][4];
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const float x0 = j == 0 ? vd[r].x : j == 1 ? vd[r].y : j == 2 ? vd[r].z : vd[r].w;
                    const float x1 = j == 0 ? vd[r + 1].x : j == 1 ? vd[r + 1].y : j == 2 ? vd[r + 1].z : vd[r + 1].w;
                    split2_bf16x3(x0, x1, p[0][r / 2], p[1][r / 2], p[2][r / 2]);
                }
                const int n = 4 * d_nc + j;
                uint16_t* d = dzt + n * CW_PP + ((d_o & ~3) | ((d_o & 3) ^ (d_nc >> 1))) * 8;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<u32x4v*>(d + pl * CW_PLANE) = u32x4v{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { bias4.x += vd[r].x; bias4.y += vd[r].y; bias4.z += vd[r].z; bias4.w += vd[r].w; }
        }
    };

    f32x16 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // ---- MFMA roles: lane (i = kx*4 + c, g): patch column (kx, c), pixel octet 2 blk + g of the block.  8 consecutive output
    // pixels are 8 consecutive bf16 of T (+1 element for kx >= 4: dword reads from the aligned start, v_alignbyte by 2 bytes).
    const int kx = i >> 2, cc = i & 3, shb = (kx >> 2) * 2;
    const uint8_t* tl = T + ((kx & 3) * 4 + cc) * (CW_XI * 2) + (NA * kg) * CW_TROW;
    const uint8_t* bl = reinterpret_cast<const uint8_t*>(dzt + i * CW_PP);
    const int bsw = i >> 3;
    const int blk0 = pg == 0 ? 0 : NB0 + NB1 * (pg - 1), nblk = pg == 0 ? NB0 : NB1;
    // per-block LDS offsets of this lane, image-independent: first / second half of the octet (bytes into T, packed 16 + 16
    // bits) and the swizzled octet of dzT
    uint32_t offT[NB0], offD[NB0];
#pragma unroll
    for (int q = 0; q < NB0; ++q) {
        const int oo = 2 * (blk0 + min(q, nblk - 1)) + g;
        const int p0 = 8 * oo, p1 = p0 + 4;
        const int oyA = p0 / C1_OW, oxA = p0 - oyA * C1_OW, oyB = p1 / C1_OW, oxB = p1 - oyB * C1_OW;
        offT[q] = (uint32_t)(oyA * (C1_S * CW_TROW) + 2 * oxA) | ((uint32_t)(oyB * (C1_S * CW_TROW) + 2 * oxB) << 16);
        offD[q] = (uint32_t)(((oo & ~3) | ((oo & 3) ^ bsw)) * 16);
    }

    // the next image's loads are issued as soon as the registers they land in have been staged (not after the barrier): the
    // dz staging pass and the barrier are part of the latency cover; the row index of the image after that is already here
    int b = blockIdx.x;
    long row_next = image_row(b + gridDim.x);
    if (b < B && !(DBG & 4)) { fetch_img(image_row(b)); fetch_dz(b); }
    for (; b < B; b += gridDim.x) {
        const bool more = b + (int)gridDim.x < B && !(DBG & 4);
        __syncthreads();                                   // previous image fully consumed
        if (!(DBG & 2)) stage_img();
        if (more) fetch_img(row_next);
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & 2)) stage_dz();
        if (more) fetch_dz(b + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        row_next = image_row(b + 2 * gridDim.x);
        if (DBG & 1) continue;
        __builtin_amdgcn_sched_barrier(0);
        uint32_t raw[2][NA][6];
        u32x4v bfr[2][3];
        auto rd = [&](int q, uint32_t (&rw)[NA][6], u32x4v (&bb)[3]) {
            const uint8_t* pa = tl + (offT[q] & 0xffffu);
            const uint8_t* pb = tl + (offT[q] >> 16);
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const uint2 v0 = *reinterpret_cast<const uint2*>(pa + a * CW_TROW);
                const uint2 v1 = *reinterpret_cast<const uint2*>(pb + a * CW_TROW);
                rw[a][0] = v0.x; rw[a][1] = v0.y; rw[a][2] = *reinterpret_cast<const uint32_t*>(pa + a * CW_TROW + 8);
                rw[a][3] = v1.x; rw[a][4] = v1.y; rw[a][5] = *reinterpret_cast<const uint32_t*>(pb + a * CW_TROW + 8);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bb[pl] = *reinterpret_cast<const u32x4v*>(bl + offD[q] + pl * (CW_PLANE * 2));
        };
        rd(0, raw[0], bfr[0]);
#pragma unroll
        for (int q = 0; q < NB0; ++q) {
            if (q < nblk) {
                const int cur = q & 1;
                if (q + 1 < nblk) rd(q + 1 < NB0 ? q + 1 : NB0 - 1, raw[cur ^ 1], bfr[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const u32x4v av = u32x4v{__builtin_amdgcn_alignbyte(raw[cur][a][1], raw[cur][a][0], shb),
                                             __builtin_amdgcn_alignbyte(raw[cur][a][2], raw[cur][a][1], shb),
                                             __builtin_amdgcn_alignbyte(raw[cur][a][4], raw[cur][a][3], shb),
                                             __builtin_amdgcn_alignbyte(raw[cur][a][5], raw[cur][a][4], shb)};
                    const bf16x8 af = __builtin_bit_cast(bf16x8, av);
                    if (DBG & 8) {
                        acc[a][0] += (float)af[0] + (float)af[7] + __uint_as_float(bfr[cur][0][0] ^ bfr[cur][1][1] ^ bfr[cur][2][2]);
                        continue;
                    }
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bfr[cur][2]), acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bfr[cur][1]), acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bfr[cur][0]), acc[a], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- combine the 4 pixel quarters (fixed order) and write the partial slab: [K][NF] weights / 255, then [NF] bias
    __syncthreads();
    float* red = reinterpret_cast<float*>(cws);            // [(NPG - 1) * NKG waves][NA acc][16][64 lanes]
    if (pg > 0) {
        float* r0 = red + (long)((pg - 1) * NKG + kg) * (NA * 1024);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) r0[(a * 16 + r) * 64 + lane] = acc[a][r];
    }
    __syncthreads();
    const long slab = (long)C1_K * C1_NF + C1_NF;
    float* out = part + (long)blockIdx.x * slab;
    if (pg == 0) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][r];
#pragma unroll
                for (int s = 0; s < NPG - 1; ++s) v += red[(long)(s * NKG + kg) * (NA * 1024) + (a * 16 + r) * 64 + lane];
                const int m = (NA * kg + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                out[(long)m * C1_NF + i] = v / 255.f;
            }
    }
    __syncthreads();
    float4* rb4 = reinterpret_cast<float4*>(cws);
    rb4[tid] = dz_on ? bias4 : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tid < C1_NF) {
        const int nc = tid >> 2, comp = tid & 3;
        float t = 0.f;
        for (int q = nc; q < 400; q += 8) {
            const float4 v = rb4[q];
            t += comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
        }
        out[(long)C1_K * C1_NF + tid] = t;
    }
}

// ---- half-image variant: a work unit is (image, upper / lower 10 output rows = 44 image rows), a workgroup is 4 waves and
// 81.6 KB of LDS, TWO workgroups per CU: one workgroup's staging pass, barriers and load waits run under the other's MFMA
// phase (the single 512-thread workgroup above spends ~30 % of its time in neither VALU nor MFMA issue)
constexpr int CH_ROWS = 44;
constexpr int CH_TBYTES = CH_ROWS * CW_TROW;             // 35552
constexpr int CH_PP = 240;                               // bf16 per dzT row: 200 pixels + the empty octets 25 .. 27; 120 dwords = 56 mod 64
constexpr int CH_PLANE = C1_NF * CH_PP;
constexpr size_t CH_LDS = (size_t)CH_TBYTES + (size_t)3 * CH_PLANE * 2;       // 81632
constexpr int CH_NT = 256;
// LDS bank conflicts (scripts/lds_conflicts.py; measured with SQ_LDS_BANK_CONFLICT: 0.36 of the LDS cycles of a kernel whose LDS is
// busy 0.69 of the time before these two, 0.08 predicted after):
//  * the runs (x & 3, c) of an image row are 48 bytes apart, 12 dwords: the 4-byte tail reads of the 12-byte operand windows (bank =
//    dword mod 32) of the runs j and j + 8 hit the same bank; the runs 8 .. 15 are shifted by 8 bytes into the row's padding
constexpr int CH_SKEW = 8;
//  * octet slot of (filter row n, pixel octet oo) in dzT: oo with its low two bits XORed by (n >> 4) | parity(n >> 2) << 1.  The 16-byte
//    fragment reads of a 16-lane group {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} start 30 n quads apart, -2 n mod 16: the rows n and
//    n +- 8 / 24 of a group share a start and are told apart by bit 0 (one of each pair is below 16); bit 1 is constant inside a group
//    (the parity of n >> 2 is what distinguishes the two groups), so it cannot make neighbours collide, and it spreads the 8 rows
//    4 d + j that one staging store covers over four quads instead of two.  (n >> 3, the previous choice, made the reads 2-way.)
__device__ __forceinline__ int ch_dz_swz(int n4 /* n >> 2 */) { return (n4 >> 2) | (((n4 ^ (n4 >> 1) ^ (n4 >> 2)) & 1) << 1); }
constexpr int CH_NB = 13;                                // blocks of 16 pixels: 25 octets, the 26th is empty

// PF2: the loads of unit u + 2 are issued while unit u is staged (two register sets, the unit loop unrolled by two): a
// unit's 40 KB have two unit times to arrive instead of one MFMA phase (~1.3 us, about the loaded HBM latency).
template <int DBG = 0, bool PF2 = false>
__global__ __launch_bounds__(CH_NT, 2) void c1wgrad_half_kernel(const uint8_t* __restrict__ obs, const int32_t* __restrict__ srow,
                                                                const float* __restrict__ dz, int B, float* __restrict__ part, int dither) {
    // dither (x6_dither, wres.hip.h): every other workgroup stages dz negated and writes its slab with the sign undone
    const bool sg_odd = (dither & 1) && (blockIdx.x & 1);
    const uint32_t sg_k = sg_odd ? 0x80008000u : 0x8000u;
    const float sg_s = sg_odd ? -1.f : 1.f;
    extern __shared__ __attribute__((aligned(16))) uint8_t cws[];
    uint8_t* T = cws;                                     // bf16 [44 rows][x & 3][c][x >> 2]
    uint16_t* dzt = reinterpret_cast<uint16_t*>(cws + CH_TBYTES);
    const int tid = threadIdx.x, lane = tid & 63, kg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    constexpr int NA = 2;                                 // patch rows 2kg, 2kg + 1 per wave

    // ---- staging roles
    const bool img_on = tid < 220;
    const int iy = img_on ? tid / 5 : 0, iq = img_on ? tid % 5 : 0;
    const bool img_tail = img_on && iq == 4;              // also the last 4 pixels of the row (run index 20)
    const int img_off = (iy * C1_W + 16 * iq) * C1_C;
    const bool dz_on = tid < 200;
    const int d_o = dz_on ? tid >> 3 : 0, d_nc = tid & 7;
    u32x4v viA[5], viB[PF2 ? 5 : 1];
    float4 vdA[8], vdB[PF2 ? 8 : 1];
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch_img = [&](u32x4v* vi, long row, int half) {
        const uint8_t* gi = obs + row * C1_IMG + half * (40 * C1_W * C1_C) + img_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) vi[j] = *reinterpret_cast<const u32x4v*>(gi + 16 * j);
        if (img_tail) vi[4] = *reinterpret_cast<const u32x4v*>(gi + 64);
    };
    auto fetch_dz = [&](float4* vd, int b, int half) {
        const float* gd = dz + ((long)b * C1_PIX + half * 200 + 8 * d_o) * C1_NF + 4 * d_nc;
#pragma unroll
        for (int r = 0; r < 8; ++r) vd[r] = *reinterpret_cast<const float4*>(gd + r * C1_NF);
    };
    auto image_row = [&](int u) { return (u >> 1) < B ? (srow ? (long)srow[u >> 1] : (long)(u >> 1)) : 0L; };
    auto stage_img = [&](const u32x4v* vi) {
        if (img_on && !(DBG & 16)) {
            uint8_t* d = T + iy * CW_TROW + 8 * iq;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t f0 = __float_as_uint((float)((vi[0][p] >> (8 * c)) & 0xff));
                    const uint32_t f1 = __float_as_uint((float)((vi[1][p] >> (8 * c)) & 0xff));
                    const uint32_t f2 = __float_as_uint((float)((vi[2][p] >> (8 * c)) & 0xff));
                    const uint32_t f3 = __float_as_uint((float)((vi[3][p] >> (8 * c)) & 0xff));
                    *reinterpret_cast<uint2*>(d + (p * 4 + c) * (CW_XI * 2) + CH_SKEW * (p >> 1)) =
                        make_uint2(__builtin_amdgcn_perm(f1, f0, 0x07060302u), __builtin_amdgcn_perm(f3, f2, 0x07060302u));
                }
            if (img_tail) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        *reinterpret_cast<uint16_t*>(d + 8 + (p * 4 + c) * (CW_XI * 2) + CH_SKEW * (p >> 1)) =
                            (uint16_t)(__float_as_uint((float)((vi[4][p] >> (8 * c)) & 0xff)) >> 16);
            }
        }
    };
    auto stage_dz = [&](const float4* vd) {
        if (dz_on && !(DBG & 32)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t p[3][4];
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const float x0 = j == 0 ? vd[r].x : j == 1 ? vd[r].y : j == 2 ? vd[r].z : vd[r].w;
                    const float x1 = j == 0 ? vd[r + 1].x : j == 1 ? vd[r + 1].y : j == 2 ? vd[r + 1].z : vd[r + 1].w;
                    split2_bf16x3_sg(x0, x1, sg_k, sg_s, p[0][r / 2], p[1][r / 2], p[2][r / 2]);
                }
                const int n = 4 * d_nc + j;
                uint16_t* d = dzt + n * CH_PP + ((d_o & ~3) | ((d_o & 3) ^ ch_dz_swz(d_nc))) * 8;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<u32x4v*>(d + pl * CH_PLANE) = u32x4v{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { bias4.x += vd[r].x; bias4.y += vd[r].y; bias4.z += vd[r].z; bias4.w += vd[r].w; }
        }
    };
    // the octet slots 24 .. 27 of every dzT row: one of them is rewritten with octet 24 per image, the other three stay zero
    // and are what the empty 26th octet of the last block reads
    for (int e = tid; e < 3 * C1_NF * 16; e += CH_NT) {
        const int rowi = e >> 4, w2 = e & 15;
        reinterpret_cast<uint32_t*>(dzt + rowi * CH_PP + 24 * 8)[w2] = 0u;
    }

    f32x16 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int kx = i >> 2, cc = i & 3, shb = (kx >> 2) * 2;
    const uint8_t* tl = T + ((kx & 3) * 4 + cc) * (CW_XI * 2) + CH_SKEW * ((kx & 3) >> 1) + (NA * kg) * CW_TROW;
    const uint8_t* bl = reinterpret_cast<const uint8_t*>(dzt + i * CH_PP);
    const int bsw = ch_dz_swz(i >> 2);
    uint32_t offT[CH_NB], offD[CH_NB];
#pragma unroll
    for (int q = 0; q < CH_NB; ++q) {
        const int oo = 2 * q + g;                          // 0 .. 25; octet 25 does not exist: its A reads are clamped to octet 24
        const int oa = min(oo, 24);                        // (finite values), its dz slot is zero
        const int p0 = 8 * oa, p1 = p0 + 4;
        const int oyA = p0 / C1_OW, oxA = p0 - oyA * C1_OW, oyB = p1 / C1_OW, oxB = p1 - oyB * C1_OW;
        offT[q] = (uint32_t)(oyA * (C1_S * CW_TROW) + 2 * oxA) | ((uint32_t)(oyB * (C1_S * CW_TROW) + 2 * oxB) << 16);
        offD[q] = (uint32_t)(((oo & ~3) | ((oo & 3) ^ bsw)) * 16);
    }

    const int nunits = 2 * B;
    const int G = gridDim.x;
    auto mfma_phase = [&]() {
        __builtin_amdgcn_s_setprio(1);                     // the MFMA phase outranks the co-resident workgroup's staging pass
        __builtin_amdgcn_sched_barrier(0);
        uint32_t raw[2][NA][6];
        u32x4v bfr[2][3];
        auto rd = [&](int q, uint32_t (&rw)[NA][6], u32x4v (&bb)[3]) {
            const uint8_t* pa = tl + (offT[q] & 0xffffu);
            const uint8_t* pb = tl + (offT[q] >> 16);
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const uint2 v0 = *reinterpret_cast<const uint2*>(pa + a * CW_TROW);
                const uint2 v1 = *reinterpret_cast<const uint2*>(pb + a * CW_TROW);
                rw[a][0] = v0.x; rw[a][1] = v0.y; rw[a][2] = *reinterpret_cast<const uint32_t*>(pa + a * CW_TROW + 8);
                rw[a][3] = v1.x; rw[a][4] = v1.y; rw[a][5] = *reinterpret_cast<const uint32_t*>(pb + a * CW_TROW + 8);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bb[pl] = *reinterpret_cast<const u32x4v*>(bl + offD[q] + pl * (CH_PLANE * 2));
        };
        rd(0, raw[0], bfr[0]);
#pragma unroll
        for (int q = 0; q < CH_NB; ++q) {
            const int cur = q & 1;
            if (q + 1 < CH_NB) rd(q + 1, raw[cur ^ 1], bfr[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const u32x4v av = u32x4v{__builtin_amdgcn_alignbyte(raw[cur][a][1], raw[cur][a][0], shb),
                                         __builtin_amdgcn_alignbyte(raw[cur][a][2], raw[cur][a][1], shb),
                                         __builtin_amdgcn_alignbyte(raw[cur][a][4], raw[cur][a][3], shb),
                                         __builtin_amdgcn_alignbyte(raw[cur][a][5], raw[cur][a][4], shb)};
                const bf16x8 af = __builtin_bit_cast(bf16x8, av);
                if (DBG & 8) continue;
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bfr[cur][2]), acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bfr[cur][1]), acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bfr[cur][0]), acc[a], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // one unit: stage the registers of set (vi, vd), refill them with unit `un` (clamped: past the end a valid unit is
    // re-read and never staged), multiply
    auto unit_step = [&](u32x4v* vi, float4* vd, int un) {
        const int uc = min(un, nunits - 1);
        __syncthreads();                                   // previous unit fully consumed
        if (!(DBG & 2)) stage_img(vi);
        if (!(DBG & 4)) fetch_img(vi, image_row(uc), uc & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & 2)) stage_dz(vd);
        if (!(DBG & 4)) fetch_dz(vd, uc >> 1, uc & 1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (!(DBG & 1)) mfma_phase();
    };
    int u = blockIdx.x;
    if constexpr (PF2) {
        if (u < nunits) {
            { const int uc = u; fetch_img(viA, image_row(uc), uc & 1); fetch_dz(vdA, uc >> 1, uc & 1); }
            { const int uc = min(u + G, nunits - 1); fetch_img(viB, image_row(uc), uc & 1); fetch_dz(vdB, uc >> 1, uc & 1); }
        }
        for (; u < nunits; u += 2 * G) {
            unit_step(viA, vdA, u + 2 * G);
            if (u + G < nunits) unit_step(viB, vdB, u + 3 * G);
        }
    } else {
        if (u < nunits && !(DBG & 4)) { fetch_img(viA, image_row(u), u & 1); fetch_dz(vdA, u >> 1, u & 1); }
        for (; u < nunits; u += G) unit_step(viA, vdA, u + G);
    }

    // ---- partial slab of this workgroup: [K][NF] weights / 255, then [NF] bias
    const long slab = (long)C1_K * C1_NF + C1_NF;
    float* out = part + (long)blockIdx.x * slab;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (NA * kg + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            out[(long)m * C1_NF + i] = (sg_s * acc[a][r]) / 255.f;
        }
    __syncthreads();
    float4* rb4 = reinterpret_cast<float4*>(cws);
    rb4[tid] = dz_on ? bias4 : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tid < C1_NF) {
        const int nc = tid >> 2, comp = tid & 3;
        float t = 0.f;
        for (int q = nc; q < 200; q += 8) {
            const float4 v = rb4[q];
            t += comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
        }
        out[(long)C1_K * C1_NF + tid] = t;
    }
}

inline hipError_t launch_c1wgrad_half(const void* obs, const int32_t* srow, const float* dz, int B, float* part, int nblocks,
                                      hipStream_t stream, bool pf2 = false, int dbg = 0) {
    auto kern = pf2 ? c1wgrad_half_kernel<0, true> : c1wgrad_half_kernel<0, false>;
#ifdef MRL_X6_EXPERIMENTS       // phase omissions (timing only): option c1_dbg = 64 + bits (1 no MFMA phase, 2 no staging pass, 4 no loads, 8 no MFMAs, 16 no image staging, 32 no dz staging)
    switch (pf2 ? dbg : 0) {
        case 1: kern = c1wgrad_half_kernel<1, true>; break;   case 2: kern = c1wgrad_half_kernel<2, true>; break;
        case 4: kern = c1wgrad_half_kernel<4, true>; break;   case 8: kern = c1wgrad_half_kernel<8, true>; break;
        case 16: kern = c1wgrad_half_kernel<16, true>; break; case 32: kern = c1wgrad_half_kernel<32, true>; break;
        case 6: kern = c1wgrad_half_kernel<6, true>; break;   case 48: kern = c1wgrad_half_kernel<48, true>; break;
        default: break;
    }
#endif
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(CH_NT), CH_LDS, stream, static_cast<const uint8_t*>(obs), srow, dz, B, part, x6_dither());
    return hipGetLastError();
}

template <int DBG = 0>
inline hipError_t launch_c1wgrad(const void* obs, const int32_t* srow, const float* dz, int B, float* part, int nblocks,
                                 hipStream_t stream, int dbg = 0) {
#ifdef MRL_X6_EXPERIMENTS
    if (DBG == 0 && dbg) {
        switch (dbg) {
        case 1: return launch_c1wgrad<1>(obs, srow, dz, B, part, nblocks, stream);
        case 2: return launch_c1wgrad<2>(obs, srow, dz, B, part, nblocks, stream);
        case 3: return launch_c1wgrad<3>(obs, srow, dz, B, part, nblocks, stream);
        case 4: return launch_c1wgrad<4>(obs, srow, dz, B, part, nblocks, stream);
        case 6: return launch_c1wgrad<6>(obs, srow, dz, B, part, nblocks, stream);
        case 8: return launch_c1wgrad<8>(obs, srow, dz, B, part, nblocks, stream);
        case 14: return launch_c1wgrad<14>(obs, srow, dz, B, part, nblocks, stream);
        case 16: return launch_c1wgrad<16>(obs, srow, dz, B, part, nblocks, stream);
        case 32: return launch_c1wgrad<32>(obs, srow, dz, B, part, nblocks, stream);
        case 17: return launch_c1wgrad<17>(obs, srow, dz, B, part, nblocks, stream);
        case 33: return launch_c1wgrad<33>(obs, srow, dz, B, part, nblocks, stream);
        default: break;
        }
    }
#endif
    auto kern = c1wgrad_kernel<DBG>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(CW_NT), CW_LDS, stream, static_cast<const uint8_t*>(obs), srow, dz, B, part);
    return hipGetLastError();
}

}  // namespace mrl
