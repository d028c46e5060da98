// Image-resident conv weight gradient on the BF16 matrix pipe with fp32-class products, for gfx950 (MI355X):
//
//   dW[(ky,kx,c)][n] = sum over (b, oy, ox) of  X[b, oy*s+ky, ox*s+kx, c] * dz[b, oy, ox, n]
//   db[n]            = sum over (b, oy, ox) of  dz[b, oy, ox, n]
//
// (tf.gradients of a2c/utils.py:37-56 `conv`, taken by ppo2/model.py:100-109) for NatureCNN's conv2 / conv3
// (common/models.py:21-22).  Replaces the fp32-MFMA engine of imgres.hip.h for these two layers (24 % of the benched step
// at 0.72 of the fp32 pipe, which is only 0.36 of what the split arithmetic reaches on the bf16 pipe).
//
// Arithmetic: that of gemmx6.hip.h -- both operands split EXACTLY into three bf16 planes (split2_bf16x3, wres.hip.h:
// round-to-nearest at each level, all residuals exact), the six plane products x_i w_j with i + j <= 2 accumulated in fp32,
// small terms first; what is dropped is at most 2^-24 of a product, one fp32 rounding (8 products in -DMRL_PRODUCTS8 builds).
//
// The contraction index of this GEMM is the output PIXEL, the slow index of both operands in memory, while
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive contraction elements per lane.  The tiled engine (wgradx8.hip.h) transposes
// in its staging pass and re-splits every input element once per overlapping patch (4x / 9x).  Here instead:
//   * a persistent workgroup per CU stages a WHOLE image and its dz map once: coalesced 16-byte loads (the next image's
//     loads stay in flight in registers during the MFMA phase), split ONCE per element, written to LDS in their natural
//     layout [pixel][plane][channel] with 8-byte stores -- no transpose anywhere in the staging pass;
//   * the transpose happens in the LDS READ: `ds_read_b64_tr_b16` hands lane (column c, k group) four consecutive
//     contraction elements of its column from a [4 k][16 c] block whose four k rows are FOUR INDEPENDENT ADDRESSES --
//     so the four k rows of a block are simply the addresses of four patch pixels, and im2col (stride, tap offset, row
//     wrap of the output map) costs nothing: per-lane pixel offsets are computed once, taps / planes / channel blocks
//     are instruction immediates.  No padded "garbage" columns: the contraction runs over the NPIX real output pixels,
//     rounded up to a multiple of 16 with zero dz rows;
//   * every wave keeps its TM x TN accumulator tiles (tap x 32 channels  x  32 filters) in registers for the kernel's
//     lifetime; two barriers per image; one partial slab per workgroup, combined in fixed order by reduce_slabs.
// Measured and dropped (profiles/README.md, r03d): splitting image b+1 in registers in the MIDDLE of the MFMA phase of image b
// (only the LDS writes left between the barriers) -- 4.78 vs 4.64 ms: VALU issued beside the partner wave's MFMAs is paid in
// matrix time on this chip (the "SIMD time = MFMA + 4 x VALU" rule of DESIGN.md 3.3), and 250 instead of 170 VGPRs.
// Pixel strides are padded so that the four k rows of a transpose read fall into different quarters of the 256-byte
// bank row (scripts/tr_probe.hip measures the patterns).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hip.h"
#include "wres.hip.h"      // bf16x8, split2_bf16x3

namespace mrl {

typedef short wt_v4i16 __attribute__((ext_vector_type(4)));
typedef short wt_v8i16 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) wt_v4i16 wt_lds_v4i16;

// 4 consecutive contraction elements of this lane's column (see the header): source role of lane l inside its 16-lane
// group: k row (l & 15) >> 2, columns 4 * (l & 3) .. + 3 at the byte address it passes
__device__ __forceinline__ wt_v4i16 wt_tr_read(const uint8_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((wt_lds_v4i16*)(p));
}
__device__ __forceinline__ bf16x8 wt_frag(const wt_v4i16& lo, const wt_v4i16& hi) {
    const wt_v8i16 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int H, int W, int C, int RF, int STRIDE, int NF, int WAVES, int TM, int TN, int XPAD, int DPAD>
struct WgTrCfg {
    static constexpr int OH = (H - RF) / STRIDE + 1, OW = (W - RF) / STRIDE + 1, NPIX = OH * OW;
    static constexpr int NCH = (NPIX + 15) / 16;                     // contraction chunks of 16 pixels
    static constexpr int K = RF * RF * C, MT = K / 32, NTN = NF / 32; // 32 x 32 output tiles
    static constexpr int CB = C / 32;                                // 32-channel blocks per tap
    static constexpr int NT = WAVES * 64;
    static constexpr int XPS = 3 * C * 2 + XPAD;                     // bytes per input pixel: [plane][c] bf16 + pad
    static constexpr int DPS = 3 * NF * 2 + DPAD;                    // bytes per dz pixel:    [plane][n] bf16 + pad
    static constexpr int X_BYTES = H * W * XPS;
    static constexpr int D_BYTES = NCH * 16 * DPS;                   // rows NPIX .. NCH*16-1 stay zero
    static constexpr size_t LDS_BYTES = (size_t)X_BYTES + D_BYTES;
    static constexpr int XV = H * W * C / 4, DZV = NPIX * NF / 4;     // float4 per image / per dz map
    static constexpr int NXV = (XV + NT - 1) / NT, NDV = (DZV + NT - 1) / NT;
    static_assert(C % 32 == 0 && NF % 32 == 0 && K % 32 == 0, "32-wide MFMA tiles");
    static_assert(WAVES * TM * TN == MT * NTN, "waves x tiles must cover the output exactly");
    static_assert(NT % (C / 4) == 0 && NT % (NF / 4) == 0, "staging: a thread keeps its channel / filter quad");
    static_assert(XPS % 8 == 0 && DPS % 8 == 0 && X_BYTES % 16 == 0, "8-byte aligned transpose reads");
    static_assert(LDS_BYTES <= 160 * 1024, "image + dz planes must fit the CU's LDS");
    static_assert((RF - 1) * (W + 1) * XPS + 3 * C * 2 < 65536, "tap offsets must fit the DS immediate");
};

// Wave -> tiles.  The K / 32 m tiles are numbered t = (ky * CB + cb) * RF + kx (CB = C / 32 channel blocks); a wave owns TM
// consecutive ones -- TM taps kx .. kx + TM - 1 of ONE kernel row and ONE channel block, so their LDS addresses differ by
// immediates (one pixel stride each) -- and TN consecutive n tiles.
// PIPE (round 5, option wgrad_tr = 1): the split arithmetic of image b+1 runs BETWEEN the MFMAs of image b's later chunks and its
// planes wait in registers for the barrier (only the LDS stores are left between the barriers).  All waves of the one workgroup
// per CU walk their phases together, so a split phase of its own keeps the matrix pipe idle for its whole length (fake-split
// bound: -17 % of the kernel); placed instruction by instruction behind MFMAs most of it hides (profiles/r05b_interleave_ubench.txt:
// 1 wave per SIMD, 48 MFMAs + 320 VALU: 5.5 ms as two blocks, 3.7 ms interleaved).  Round 3 tried the same split as ONE block in
// the middle of the MFMA phase -- slower; the difference is the placement.  Same products, same order: bit-identical.
template <int H, int W, int C, int RF, int STRIDE, int NF, int WAVES, int TM, int TN, int XPAD, int DPAD, int DBG = 0, bool PIPE = false>
__global__ __launch_bounds__(WAVES * 64) void wgrad_tr_kernel(const float* __restrict__ x, const float* __restrict__ dz, int B,
                                                              float* __restrict__ part, int dither) {
    using G = WgTrCfg<H, W, C, RF, STRIDE, NF, WAVES, TM, TN, XPAD, DPAD>;
    static_assert(RF % TM == 0, "the m tiles of a wave are taps of one kernel row");
    extern __shared__ __attribute__((aligned(16))) uint8_t wt_lds[];
    uint8_t* xs = wt_lds;
    uint8_t* ds = wt_lds + G::X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NGRP = G::NTN / TN;
    const int mt0 = (wave / NGRP) * TM, nt0 = (wave % NGRP) * TN;

    // dither (x6_dither, wres.hip.h): every other workgroup stages dz NEGATED (no extra instruction) and writes its partial slab with
    // the sign undone -- the matrix instruction's bias toward -inf then has opposite signs in neighbouring slabs
    // and cancels in reduce_slabs' sum instead of adding up over the 256 slabs
    const bool sg_odd = dither && (blockIdx.x & 1);              // the launcher passes bit 0 of x6_dither
    const uint32_t sg_k = sg_odd ? 0x80008000u : 0x8000u;
    const float sg_s = sg_odd ? -1.f : 1.f;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 bias4 = f4zero();

    // zero rows of the dz planes behind the map (contraction padding), once
    for (int e = tid; e < (G::NCH * 16 - G::NPIX) * (G::DPS / 8); e += G::NT)
        reinterpret_cast<uint2*>(ds + G::NPIX * G::DPS)[e] = make_uint2(0u, 0u);

    // ---- per-lane operand addresses of the transpose reads
    // lane l: 16-lane group g = l >> 4 -> 16-column half mb = g & 1 of the 32-wide tile, k half hh = g >> 1;
    // source role p = l & 15 -> k row kr = p >> 2, column quad cq = p & 3.  Read tq (0, 1) of chunk q covers
    // contraction elements 16 q + 8 hh + 4 tq + kr.
    const int g = lane >> 4, mb = g & 1, hh = g >> 1, p = lane & 15, kr = p >> 2, cq = p & 3;
    const int colb = (16 * mb + 4 * cq) * 2;                                    // byte offset of the column quad
    // m tiles mt0 .. mt0 + TM - 1: kernel row ky, channel block cb, taps kx0 .. kx0 + TM - 1
    const int ky = mt0 / (G::CB * RF), cb = (mt0 / RF) % G::CB, kx0 = mt0 % RF;
    const int xbase = (ky * W + kx0) * G::XPS + cb * 64 + colb;
    int xoff[G::NCH][2];
#pragma unroll
    for (int q = 0; q < G::NCH; ++q)
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            const int r = 16 * q + 8 * hh + 4 * tq + kr;
            const int rr = r < G::NPIX ? r : 0;                                 // padding rows: any valid pixel (dz is zero)
            const int oy = rr / G::OW, ox = rr - oy * G::OW;
            xoff[q][tq] = xbase + (oy * STRIDE * W + ox * STRIDE) * G::XPS;
        }
    const int doff = (8 * hh + kr) * G::DPS + nt0 * 64 + colb;                  // + (16 q + 4 tq) * DPS, + plane, + n tile
    constexpr int A_STEP = G::XPS;                                              // next tap of the kernel row: one pixel

    // ---- staging: global -> registers (16-byte loads) -> split -> LDS [pixel][plane][c]
    // (C = 32: the 16 lanes of an 8-byte LDS store group hold two pixels whose bank windows overlap on 8 of 32 banks with the
    // 224-byte pixel pitch; taking the pixels of a group of four in the order 0 2 1 3 removes those conflicts -- SQ_LDS_BANK_CONFLICT
    // 0.23 -> 0.10 of the LDS cycles -- without making the kernel faster (same-box A/B: 4.13 / 4.13 ms).  Not used.)
    float4 rx[G::NXV], rd[G::NDV];
    auto issue_loads = [&](int bb) {
        const float4* gx = reinterpret_cast<const float4*>(x + (long)bb * (H * W * C));
        const float4* gd = reinterpret_cast<const float4*>(dz + (long)bb * (G::NPIX * NF));
#pragma unroll
        for (int q = 0; q < G::NXV; ++q) { const int e = tid + q * G::NT; rx[q] = gx[e < G::XV ? e : G::XV - 1]; }
#pragma unroll
        for (int q = 0; q < G::NDV; ++q) { const int e = tid + q * G::NT; rd[q] = gd[e < G::DZV ? e : G::DZV - 1]; }
    };
    constexpr int XQ = C / 4, DQ = NF / 4;                     // float4 per pixel
    const int xw0 = (tid / XQ) * G::XPS + (tid % XQ) * 8;      // + q * (NT / XQ) * XPS
    const int dw0 = (tid / DQ) * G::DPS + (tid % DQ) * 8;
    auto write_stage = [&]() {
#pragma unroll
        for (int q = 0; q < G::NXV; ++q) {
            const int e = tid + q * G::NT;
            if (e < G::XV) {
                uint32_t a0x, a1x, a2x, a0y, a1y, a2y;
                split2_bf16x3(rx[q].x, rx[q].y, a0x, a1x, a2x);
                split2_bf16x3(rx[q].z, rx[q].w, a0y, a1y, a2y);
                uint8_t* d = xs + xw0 + q * (G::NT / XQ) * G::XPS;
                *reinterpret_cast<uint2*>(d) = make_uint2(a0x, a0y);
                *reinterpret_cast<uint2*>(d + C * 2) = make_uint2(a1x, a1y);
                *reinterpret_cast<uint2*>(d + C * 4) = make_uint2(a2x, a2y);
            }
        }
#pragma unroll
        for (int q = 0; q < G::NDV; ++q) {
            const int e = tid + q * G::NT;
            if (e < G::DZV) {
                uint32_t a0x, a1x, a2x, a0y, a1y, a2y;
                split2_bf16x3_sg(rd[q].x, rd[q].y, sg_k, sg_s, a0x, a1x, a2x);
                split2_bf16x3_sg(rd[q].z, rd[q].w, sg_k, sg_s, a0y, a1y, a2y);
                uint8_t* d = ds + dw0 + q * (G::NT / DQ) * G::DPS;
                *reinterpret_cast<uint2*>(d) = make_uint2(a0x, a0y);
                *reinterpret_cast<uint2*>(d + NF * 2) = make_uint2(a1x, a1y);
                *reinterpret_cast<uint2*>(d + NF * 4) = make_uint2(a2x, a2y);
                bias4.x += rd[q].x; bias4.y += rd[q].y; bias4.z += rd[q].z; bias4.w += rd[q].w;
            }
        }
    };

    // PIPE: the planes of the next image, split ahead of the barrier.  (u32x4v = {plane 0 | 1 | 2}[q] as {xy, zw} pairs)
    uint32_t px[PIPE ? G::NXV : 1][6], pd[PIPE ? G::NDV : 1][6];
    bool split_live = true;                            // false while the (already split) last image of this workgroup is multiplied
    auto split_x = [&](int q) {
        split2_bf16x3(rx[q].x, rx[q].y, px[q][0], px[q][2], px[q][4]);
        split2_bf16x3(rx[q].z, rx[q].w, px[q][1], px[q][3], px[q][5]);
    };
    auto split_d = [&](int q) {
        split2_bf16x3_sg(rd[q].x, rd[q].y, sg_k, sg_s, pd[q][0], pd[q][2], pd[q][4]);
        split2_bf16x3_sg(rd[q].z, rd[q].w, sg_k, sg_s, pd[q][1], pd[q][3], pd[q][5]);
        if (split_live && tid + q * G::NT < G::DZV) { bias4.x += rd[q].x; bias4.y += rd[q].y; bias4.z += rd[q].z; bias4.w += rd[q].w; }
    };
    auto write_planes = [&]() {
#pragma unroll
        for (int q = 0; q < G::NXV; ++q)
            if (tid + q * G::NT < G::XV) {
                uint8_t* d = xs + xw0 + q * (G::NT / XQ) * G::XPS;
                *reinterpret_cast<uint2*>(d) = make_uint2(px[q][0], px[q][1]);
                *reinterpret_cast<uint2*>(d + C * 2) = make_uint2(px[q][2], px[q][3]);
                *reinterpret_cast<uint2*>(d + C * 4) = make_uint2(px[q][4], px[q][5]);
            }
#pragma unroll
        for (int q = 0; q < G::NDV; ++q)
            if (tid + q * G::NT < G::DZV) {
                uint8_t* d = ds + dw0 + q * (G::NT / DQ) * G::DPS;
                *reinterpret_cast<uint2*>(d) = make_uint2(pd[q][0], pd[q][1]);
                *reinterpret_cast<uint2*>(d + NF * 2) = make_uint2(pd[q][2], pd[q][3]);
                *reinterpret_cast<uint2*>(d + NF * 4) = make_uint2(pd[q][4], pd[q][5]);
            }
    };
    // the split of the next image is spread over the LAST SPLIT_CH chunks of the MFMA phase (its loads were issued at the phase's
    // start and have that long to arrive): pieces [lo, hi) of the NXV + NDV staged float4s go behind chunk q's MFMAs
    constexpr int NPIECE = G::NXV + G::NDV;
    constexpr int SPLIT_CH = G::NCH >= 4 ? G::NCH - 2 : G::NCH - 1;
    constexpr int MF_CH = TM * TN * kSplitProducts;                 // MFMAs per chunk and wave

    int b = blockIdx.x;
    if (b < B) issue_loads(b);
    if constexpr (PIPE) {
        if (b < B) {                                       // the first image: split in the open
#pragma unroll
            for (int q = 0; q < G::NXV; ++q) split_x(q);
#pragma unroll
            for (int q = 0; q < G::NDV; ++q) split_d(q);
        }
    }
    for (; b < B; b += gridDim.x) {
        __syncthreads();                                   // the previous image's fragment reads are done
        if constexpr (PIPE) write_planes();
        else if (!(DBG & 1) || b == (int)blockIdx.x) write_stage();      // DBG 1 (timing experiment): stage the first image only
        const int bn = b + gridDim.x;
        if (bn < B) issue_loads(bn);                       // in flight during the MFMA phase
        split_live = bn < B;
        __syncthreads();
        // ---- MFMA phase: NCH chunks of 16 pixels
        if (DBG & 2) continue;                             // DBG 2 (timing experiment): no MFMA phase
#pragma unroll
        for (int q = 0; q < G::NCH; ++q) {
            if constexpr (PIPE) {
                // (the workgroup's last image: the split runs on stale registers and is never written)
                constexpr int q0 = G::NCH - SPLIT_CH;
                if (q >= q0) {
                    const int lo = (q - q0) * NPIECE / SPLIT_CH, hi = (q - q0 + 1) * NPIECE / SPLIT_CH;
#pragma unroll
                    for (int e = 0; e < NPIECE; ++e)
                        if (e >= lo && e < hi) { if (e < G::NXV) split_x(e); else split_d(e - G::NXV); }
                }
            }
            bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const wt_v4i16 lo = wt_tr_read(xs + xoff[q][0] + a * A_STEP + pl * (C * 2));
                    const wt_v4i16 hi = wt_tr_read(xs + xoff[q][1] + a * A_STEP + pl * (C * 2));
                    fa[a][pl] = wt_frag(lo, hi);
                }
#pragma unroll
            for (int c = 0; c < TN; ++c)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const wt_v4i16 lo = wt_tr_read(ds + doff + (16 * q) * G::DPS + c * 64 + pl * (NF * 2));
                    const wt_v4i16 hi = wt_tr_read(ds + doff + (16 * q + 4) * G::DPS + c * 64 + pl * (NF * 2));
                    fb[c][pl] = wt_frag(lo, hi);
                }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int c = 0; c < TN; ++c) {      // 8 of the 9 partial products, small terms first (gemmx6.hip.h)
                    if (kCross21) {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[c][1], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[c][2], acc[a][c], 0, 0, 0);
                    }
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[c][0], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[c][1], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[c][2], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[c][0], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[c][1], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[c][0], acc[a][c], 0, 0, 0);
                }
            if constexpr (PIPE) {
                constexpr int q0 = G::NCH - SPLIT_CH;
                if (q >= q0) {
                    // this chunk's fragment reads first, then one MFMA : VPM split instructions (30 + moves per piece)
                    constexpr int VPM = (NPIECE * 34 / SPLIT_CH + MF_CH - 1) / MF_CH;
                    __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN) * 3 * 2, 0);
#pragma unroll
                    for (int m = 0; m < MF_CH; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- partial slab of this workgroup: [K][NF] weights then [NF] bias
    // C/D layout of the 32x32 MFMA: column (filter) = lane & 31, row (m) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const long slab = (long)G::K * NF + NF;
    float* out = part + (long)blockIdx.x * slab;
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int c = 0; c < TN; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = ((ky * RF + kx0 + a) * G::CB + cb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;   // K index (ky, kx, c)
                const int n = (nt0 + c) * 32 + i;
                out[(long)m * NF + n] = sg_s * acc[a][c][r];
            }
    // bias: thread t owns filter columns (4t % NF .. +3); combine the NT*4/NF threads of a column in fixed order
    float4* red = reinterpret_cast<float4*>(wt_lds);
    __syncthreads();
    red[tid] = bias4;
    __syncthreads();
    if (tid < NF) {
        constexpr int GROUPS = NF / 4;
        const int gq = tid / 4, comp = tid % 4;
        float t = 0.f;
        for (int q = gq; q < G::NT; q += GROUPS) {
            const float4 v = red[q];
            t += comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
        }
        out[(long)G::K * NF + tid] = t;
    }
}

inline int& wgrad_tr_pipe() { static int p = getenv("MRL_WGRAD_PIPE") ? atoi(getenv("MRL_WGRAD_PIPE")) : 1; return p; }   // mrl_set_option "wgrad_pipe"
template <int H, int W, int C, int RF, int STRIDE, int NF, int WAVES, int TM, int TN, int XPAD, int DPAD, int DBG = 0>
inline hipError_t launch_wgrad_tr(const float* x, const float* dz, int B, float* part, int nblocks, hipStream_t stream) {
    using G = WgTrCfg<H, W, C, RF, STRIDE, NF, WAVES, TM, TN, XPAD, DPAD>;
    const bool pipe = DBG == 0 && wgrad_tr_pipe() != 0;
    auto kern = pipe ? wgrad_tr_kernel<H, W, C, RF, STRIDE, NF, WAVES, TM, TN, XPAD, DPAD, DBG, DBG == 0>
                     : wgrad_tr_kernel<H, W, C, RF, STRIDE, NF, WAVES, TM, TN, XPAD, DPAD, DBG, false>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(G::NT), G::LDS_BYTES, stream, x, dz, B, part, x6_dither() & 1);
    return hipGetLastError();
}


// =====================================================================================================================
// Dense weight gradient (fully connected layers) with the same ingredients:
//   dW[k][n] = sum_m A[m][k] * dZ[m][n],  db[n] = sum_m dZ[m][n]        (NatureCNN fc1: K = 3136, N = 512, m = sample)
// replaces the transposed-staging tiles of wgradx8.hip.h (fc1.wgrad 3.7 ms at 0.36 of its pipe; 43 % MFMA-busy, more LDS
// bank-conflict cycles than half its LDS-active cycles).  Both operands are staged in their NATURAL layout
// [row m][plane][column] (coalesced 16-byte loads, split while staged, 8-byte LDS stores, no transposing writes) and the
// contraction-major fragments come out of `ds_read_b64_tr_b16`, whose four k rows are four consecutive samples.
//   * tile = (MT * 32) x 256 outputs per workgroup of 8 waves, wave w = 32-filter column block w x all MT row blocks:
//     a 224 x 256 tile (MT = 7) divides fc1 exactly (14 x 2 tiles), splits each A element twice and each dZ element 14
//     times (128 x 128 tiles: 4 and 25 times) -- 1.4 instead of 2.75 split instructions per MFMA;
//   * split-K over samples: `nslab` row ranges, one partial slab each (tiles of a slab fill it completely), combined in
//     fixed order by reduce_slabs; 32 samples per step, the next step's loads in flight in registers during the MFMAs;
//   * the bias column sums ride on the dZ staging registers of the workgroups of row-tile 0.
template <int MT>
struct WgTrDenseCfg {
    static constexpr int BK = MT * 32, BN = 256, R = 32, NT = 512;
    // LDS bytes per sample row: [plane][column] + pad so that consecutive rows start 16 dwords apart mod 64: the four sample rows of a
    // transpose read (64 bytes each) and the two rows a staging store group can straddle then use disjoint banks.  MT = 7: 336
    // dwords = 16 mod 64 without padding (the fixed 64-byte pad of round 3 made it 32: two-way conflicts, scripts/lds_conflicts.py)
    static constexpr int row_pad(int bytes) { return ((16 - (bytes / 4) % 64 + 64) % 64) * 4; }
    static constexpr int ARS = 3 * BK * 2 + row_pad(3 * BK * 2), BRS = 3 * BN * 2 + row_pad(3 * BN * 2);
    static constexpr int A_BYTES = R * ARS, B_BYTES = R * BRS;
    static constexpr size_t LDS_BYTES = (size_t)A_BYTES + B_BYTES;
    static constexpr int AQ = BK / 4, BQ = BN / 4;                           // float4 per row
    static constexpr int AV = R * AQ, BV = R * BQ;
    static constexpr int NAV = (AV + NT - 1) / NT, NBV = BV / NT;
    static_assert(BV % NT == 0 && NT % BQ == 0, "dZ staging: a thread keeps its filter quad");
    static_assert(LDS_BYTES <= 160 * 1024, "one stage must fit the CU's LDS");
};

template <int MT>
__global__ __launch_bounds__(512) void wgrad_tr_dense_kernel(const float* __restrict__ A, long lda, const float* __restrict__ dz,
                                                             float* __restrict__ part, long slab, int M, int K, int N,
                                                             int ktiles, int ntiles, int rows_per_slab, int dither, int per_xcd, int nblocks) {
    using G = WgTrDenseCfg<MT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t wt_lds[];
    uint8_t* as = wt_lds;
    uint8_t* bs = wt_lds + G::A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = ktiles * ntiles;
    // round 6: XCD-aware order.  Workgroup b runs on XCD b & 7; XCD x takes the contiguous run [x * per_xcd, (x + 1) * per_xcd) of the
    // logical (slab, tile) list, so a slab's T tiles -- which read the SAME dz rows -- sit on one or two XCDs instead of all eight, and the
    // ntiles workgroups that read the same A columns are neighbours in one L2 (per_xcd is a multiple of ntiles).  fc1: 252 workgroups;
    // counter traffic 4.42 -> see profiles/README.md round 6.  xcd_order = 0: blockIdx order (round 3).
    int lid = blockIdx.x;
    if (per_xcd > 0) {
        lid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= per_xcd || lid >= nblocks) return;
    }
    const int s = lid / T, tile = lid - s * T;
    const int kt = tile / ntiles, nt = tile - kt * ntiles;
    const int k0 = kt * G::BK, n0 = nt * G::BN;
    const long m_begin = (long)s * rows_per_slab;
    const long m_end = min((long)M, m_begin + rows_per_slab);
    const int nsteps = (int)((m_end - m_begin + G::R - 1) / G::R);
    const bool sg_odd = (dither & 1) && (s & 1);               // every other slab: dz staged negated, the slab written with the sign undone
    const uint32_t sg_k = sg_odd ? 0x80008000u : 0x8000u;
    const float sg_s = sg_odd ? -1.f : 1.f;

    f32x16 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float4 bias4 = f4zero();

    // transpose-read roles (see wgrad_tr_kernel): the four k rows of a read are four consecutive samples
    const int g = lane >> 4, mb = g & 1, hh = g >> 1, p = lane & 15, kr = p >> 2, cq = p & 3;
    const uint8_t* abase = as + (8 * hh + kr) * G::ARS + (16 * mb + 4 * cq) * 2;
    const uint8_t* bbase = bs + (8 * hh + kr) * G::BRS + (16 * mb + 4 * cq) * 2 + wave * 64;

    // staging roles: A float4 e = tid + q * NT -> (row e / AQ, quad e % AQ); dZ: row (tid / BQ) + q * (NT / BQ), quad tid % BQ
    int arow[G::NAV], aoff[G::NAV];            // sample row inside the step, element offset inside the A row
#pragma unroll
    for (int q = 0; q < G::NAV; ++q) {
        const int e = tid + q * G::NT, ec = e < G::AV ? e : G::AV - 1;
        arow[q] = ec / G::AQ;
        aoff[q] = (ec - arow[q] * G::AQ) * 4;
    }
    const int brow = tid / G::BQ, boff = (tid % G::BQ) * 4;
    float4 ra[G::NAV], rb[G::NBV];
    auto issue_loads = [&](int step) {
        const long m0 = m_begin + (long)step * G::R;
#pragma unroll
        for (int q = 0; q < G::NAV; ++q) {
            const long m = min(m0 + arow[q], m_end - 1);
            ra[q] = *reinterpret_cast<const float4*>(A + m * lda + k0 + aoff[q]);
        }
#pragma unroll
        for (int q = 0; q < G::NBV; ++q) {
            const long m = min(m0 + brow + q * (G::NT / G::BQ), m_end - 1);
            rb[q] = *reinterpret_cast<const float4*>(dz + m * N + n0 + boff);
        }
    };
    auto write_stage = [&](int step) {
        const long m0 = m_begin + (long)step * G::R;
#pragma unroll
        for (int q = 0; q < G::NAV; ++q) {
            if (tid + q * G::NT < G::AV) {
                float4 v = ra[q];
                if (m0 + arow[q] >= m_end) v = f4zero();                    // rows behind the range contribute nothing
                uint32_t a0x, a1x, a2x, a0y, a1y, a2y;
                split2_bf16x3(v.x, v.y, a0x, a1x, a2x);
                split2_bf16x3(v.z, v.w, a0y, a1y, a2y);
                uint8_t* d = as + arow[q] * G::ARS + aoff[q] * 2;
                *reinterpret_cast<uint2*>(d) = make_uint2(a0x, a0y);
                *reinterpret_cast<uint2*>(d + G::BK * 2) = make_uint2(a1x, a1y);
                *reinterpret_cast<uint2*>(d + G::BK * 4) = make_uint2(a2x, a2y);
            }
        }
#pragma unroll
        for (int q = 0; q < G::NBV; ++q) {
            float4 v = rb[q];
            const int row = brow + q * (G::NT / G::BQ);
            if (m0 + row >= m_end) v = f4zero();
            uint32_t a0x, a1x, a2x, a0y, a1y, a2y;
            split2_bf16x3_sg(v.x, v.y, sg_k, sg_s, a0x, a1x, a2x);
            split2_bf16x3_sg(v.z, v.w, sg_k, sg_s, a0y, a1y, a2y);
            uint8_t* d = bs + row * G::BRS + boff * 2;
            *reinterpret_cast<uint2*>(d) = make_uint2(a0x, a0y);
            *reinterpret_cast<uint2*>(d + G::BN * 2) = make_uint2(a1x, a1y);
            *reinterpret_cast<uint2*>(d + G::BN * 4) = make_uint2(a2x, a2y);
            bias4.x += v.x; bias4.y += v.y; bias4.z += v.z; bias4.w += v.w;
        }
    };

    if (nsteps > 0) issue_loads(0);
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();                                   // the previous step's fragment reads are done
        write_stage(step);
        if (step + 1 < nsteps) issue_loads(step + 1);      // in flight during the MFMA phase
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < G::R / 16; ++ch) {
            bf16x8 fb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const wt_v4i16 lo = wt_tr_read(bbase + (16 * ch) * G::BRS + pl * (G::BN * 2));
                const wt_v4i16 hi = wt_tr_read(bbase + (16 * ch + 4) * G::BRS + pl * (G::BN * 2));
                fb[pl] = wt_frag(lo, hi);
            }
            // explicit fragment pipeline (row block j + 1 is read while the 8 MFMAs of row block j run); the fences keep the
            // scheduler from hoisting all 7 row blocks' reads (84 VGPRs) in front of the first MFMA
            bf16x8 fa[2][3];
            auto read_a = [&](int j, bf16x8 (&f)[3]) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const wt_v4i16 lo = wt_tr_read(abase + (16 * ch) * G::ARS + pl * (G::BK * 2) + j * 64);
                    const wt_v4i16 hi = wt_tr_read(abase + (16 * ch + 4) * G::ARS + pl * (G::BK * 2) + j * 64);
                    f[pl] = wt_frag(lo, hi);
                }
            };
            read_a(0, fa[0]);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                if (j + 1 < MT) read_a(j + 1, fa[(j + 1) & 1]);
                const bf16x8(&f)[3] = fa[j & 1];
                // 8 of the 9 partial products, small terms first (gemmx6.hip.h)
                if (kCross21) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], fb[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], fb[2], acc[j], 0, 0, 0);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], fb[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], fb[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], fb[2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], fb[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], fb[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], fb[0], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- this tile's part of slab s: [K][N] weights then [N] bias
    float* out = part + (long)s * slab;
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = k0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[(long)m * N + n0 + wave * 32 + i] = sg_s * acc[j][r];
        }
    if (kt == 0) {      // bias: thread t owns filter columns n0 + 4 (t % BQ) .. + 3; the NT / BQ threads of a quad in fixed order
        float4* red = reinterpret_cast<float4*>(wt_lds);
        __syncthreads();
        red[tid] = bias4;
        __syncthreads();
        if (tid < G::BN) {
            const int quad = tid / 4, comp = tid % 4;
            float t = 0.f;
            for (int q = quad; q < G::NT; q += G::BQ) {
                const float4 v = red[q];
                t += comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
            }
            out[(long)K * N + n0 + tid] = t;
        }
    }
}

inline int& wgrad_tr_xcd() { static int p = getenv("MRL_WGRAD_XCD") ? atoi(getenv("MRL_WGRAD_XCD")) : 1; return p; }     // mrl_set_option "wgrad_xcd"
struct WgTrDensePlan { int mt = 0, ktiles = 0, ntiles = 0, nslab = 0, rows_per_slab = 0; };
// usable when the tiles divide the output exactly and the operands allow 16-byte loads
inline WgTrDensePlan wgrad_tr_dense_plan(long M, int K, int N, long lda, int num_cus, size_t part_floats) {
    WgTrDensePlan p;
    if (M < 1024 || M > 0x7fffffffL || N % 256 != 0 || lda % 4 != 0) return p;
    if (K % 224 == 0) p.mt = 7; else if (K % 256 == 0) p.mt = 8; else return p;
    p.ktiles = K / (p.mt * 32);
    p.ntiles = N / 256;
    const int T = p.ktiles * p.ntiles;
    const long slab = (long)K * N + N;
    long S = std::max<long>(1, num_cus / T);
    S = std::min<long>(S, (long)(part_floats / slab));
    if (S < 1) { p.mt = 0; return p; }
    p.rows_per_slab = (int)(((M + S - 1) / S + 31) / 32 * 32);
    p.nslab = (int)((M + p.rows_per_slab - 1) / p.rows_per_slab);
    return p;
}

inline hipError_t launch_wgrad_tr_dense(const float* A, long lda, const float* dz, float* part, long slab, int M, int K, int N,
                                        const WgTrDensePlan& p, hipStream_t stream) {
    const int nblocks = p.nslab * p.ktiles * p.ntiles;
    // per-XCD run of the logical list: a multiple of ntiles, so the workgroups that share A columns stay together
    const int per_xcd = wgrad_tr_xcd() ? ((nblocks + 7) / 8 + p.ntiles - 1) / p.ntiles * p.ntiles : 0;
    const unsigned blocks = per_xcd ? (unsigned)(8 * per_xcd) : (unsigned)nblocks;
    auto go = [&](auto kern, size_t lds) {
        { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, stream, A, lda, dz, part, slab, M, K, N, p.ktiles, p.ntiles,
                           p.rows_per_slab, x6_dither(), per_xcd, nblocks);
        return hipGetLastError();
    };
    if (p.mt == 7) return go(wgrad_tr_dense_kernel<7>, WgTrDenseCfg<7>::LDS_BYTES);
    return go(wgrad_tr_dense_kernel<8>, WgTrDenseCfg<8>::LDS_BYTES);
}

}  // namespace mrl
