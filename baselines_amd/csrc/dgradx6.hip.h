// Conv data-gradient on the split-bf16 matrix pipe, position-major tiles (gfx950).  Sixth GEMM engine of libmrl.
// Replaces the gradient of tf.nn.conv2d w.r.t. its input inside `tf.gradients(loss, params)` (ppo2/model.py:102-103)
// for the hidden conv layers of NatureCNN (common/models.py:21-22 via a2c/utils.py:37-56).
//
// Gather form, as in ldsdgrad.hip.h: input pixel (iy, ix) = (S*yy + py, S*xx + px) of stride-parity class (py, px) receives
//   dX[b, iy, ix, c] = sum_{a, b2 < TAPS} sum_n dz[b, yy - a, xx - b2, n] * W[py + S*a, px + S*b2, c, n]
// Two observations turn this into a dense GEMM without waste:
//   1. the dz pixels a class-grid position (yy, xx) reads do NOT depend on the class -- only the filter taps do.  All
//      S*S classes are therefore columns of ONE GEMM: A[row][k = (a, b2, n)] (im2col of dz over the class grid),
//      B[col = (class, c)][k] (filter taps re-ordered once per call into split bf16 planes), N = S*S*C columns
//      (conv2: 128 instead of four 32-column problems that each re-stage A), and the epilogue scatters column (class, c)
//      of row (b, yy, xx) to its unique destination pixel (deterministic, no atomics);
//   2. which taps fall outside the dz map depends only on the POSITION (yy, xx).  A workgroup tile is BM images of one
//      position, so the validity of a tap is uniform over the tile and out-of-map taps are skipped instead of being
//      multiplied by zeros: only useful MACs are executed (the LDS-resident fp32 engine spends 19 % (conv2) / 40 %
//      (conv3) of its matrix time on border zeros).
// Arithmetic, staging and tile shape are those of gemmx6.hip.h: both operands split exactly into 3 bf16 planes, 6 (or 8)
// partial products per multiply accumulated in fp32 by v_mfma_f32_32x32x16_bf16, A split while it is staged into LDS,
// 4 waves x (64 x 64) per workgroup, two workgroups per CU.  Tiles of one image group (all positions) take consecutive
// slots of ONE XCD so that the TAPS^2-fold re-reads of dz pixels by neighbouring positions hit that XCD's L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemmx6.hip.h"

namespace mrl {

inline int& dgrad_x6_pipe() { static int p = 1; return p; }      // mrl_set_option "dgrad_x6" = 2: dgrad_x6p_kernel (round 6)

template <int H, int W, int C, int RF, int S, int NF>
struct DgX6Geom {
    static constexpr int OH = (H - RF) / S + 1, OW = (W - RF) / S + 1;
    static constexpr int TAPS = (RF + S - 1) / S;
    static constexpr int HY = (H + S - 1) / S, WX = (W + S - 1) / S;
    static constexpr int NPOS = HY * WX;
    static constexpr int NCLS = S * S;
    static constexpr int N = NCLS * C;                    // GEMM columns (class, c)
    static constexpr int K = TAPS * TAPS * NF;            // GEMM k (a, b2, n)
    static constexpr int KT_PER_TAP = NF / X6_BK;
    static constexpr int NKT = K / X6_BK;
    static_assert(NF % X6_BK == 0, "a k tile lies inside one tap");
};

// B planes [plane][col = cls*C + c][k = (a*TAPS + b2)*NF + n] = split(W[py + S*a][px + S*b2][c][n]) (0 for taps beyond RF)
template <int H, int W, int C, int RF, int S, int NF>
__global__ __launch_bounds__(256) void dgx6_split_planes_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int kperm) {
    using G = DgX6Geom<H, W, C, RF, S, NF>;
    const int total = G::N * G::K;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int col = e / G::K, k = e - col * G::K;
        const int cls = col / C, c = col - cls * C;
        const int py = cls / S, px = cls - py * S;
        const int tap = k / NF, n = k - tap * NF;
        const int a = tap / G::TAPS, b2 = tap - a * G::TAPS;
        const int ky = py + S * a, kx = px + S * b2;
        const float v = (ky < RF && kx < RF) ? w[((long)(ky * RF + kx) * C + c) * NF + n] : 0.f;
        uint16_t q0, q1, q2;
        split1_bf16x3(v, q0, q1, q2);
        // kperm 1: k in the order of a plane tensor (planes.hip.h); 2: k-tile-major [k / 32][col][k % 32] -- a k tile of ALL columns is
        // contiguous, so the 16 rows x 64 bytes a wave's staging load covers are 8 whole cache lines instead of 16 half lines
        const int eo = kperm == 2 ? (k / X6_BK) * (G::N * X6_BK) + col * X6_BK + (k % X6_BK)
                     : kperm ? col * G::K + tap * NF + (int)kperm32(n) : e;
        out[0 * total + eo] = q0;
        out[1 * total + eo] = q1;
        out[2 * total + eo] = q2;
    }
}

// destination of block (a_, b_) of this lane in the transposed epilogue: element offset off_, validity ok_, first channel c0_
#define DGX6_EPI_ADDR(a_, b_, off_, ok_, c0_)                                                                  \
    const int cb_ = n0 + (wn * 2 + (b_)) * 32;                                                                 \
    const int cls_ = cb_ / C, c0_ = cb_ - cls_ * C;                                                            \
    const int py_ = cls_ / S, px_ = cls_ - py_ * S;                                                            \
    const int iy_ = yy * S + py_, ix_ = xx * S + px_;                                                          \
    const int bimg_ = b0 + (wm * 2 + (a_)) * 32 + i;                                                           \
    const bool ok_ = cb_ < G::N && iy_ < H && ix_ < W && bimg_ < B;                                            \
    const long off_ = ok_ ? (long)bimg_ * (H * W * C) + ((long)iy_ * W + ix_) * C + c0_ : 0L;

// PA: dz comes as a plane tensor (dzp = plane 0, dz_ps elements between planes; Bp laid out with kperm): staging without
//     split arithmetic.  TR: MFMA operands swapped, a lane owns one image and 16 channels of a destination pixel: the
//     epilogue reads ONE mask word per 32 channels and writes fp32 + the plane tensor of dx with 16-byte stores
//     (planes.hip.h); C % 32 == 0, ReLU below.
// IL: next tile's global loads issued between the MFMAs (see gemm_x6_kernel).
template <int H, int W, int C, int RF, int S, int NF, int WM, int WN, bool X8, bool PA = false, bool TR = false, bool IL = false>
__global__ __launch_bounds__(256, 2) void dgrad_x6_kernel(const float* __restrict__ dz, const uint16_t* __restrict__ Bp,
                                                       const float* __restrict__ hmask, const uint32_t* __restrict__ mbits,
                                                       float* __restrict__ dx, int act, int B, int btiles,
                                                       long tiles_per_xcd, long total_tiles, int slots_per_xcd, int dbg, int prio,
                                                       const uint16_t* __restrict__ dzp, long dz_ps,
                                                       uint16_t* __restrict__ dxp, long dx_ps, int dither) {
    using G = DgX6Geom<H, W, C, RF, S, NF>;
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = WM * 64, BN = WN * 64;
    static_assert(G::N % BN == 0 || G::N < BN, "column tiles");
    constexpr int NTN = (G::N + BN - 1) / BN;             // column tiles
    constexpr int NA = BM / 32, NQ = BN / 64;
    constexpr int NAP = BM / 64;                          // PA: 16-byte pieces of A per thread, plane and tile
    constexpr int OH = G::OH, OW = G::OW, TAPS = G::TAPS;
    static_assert(!TR || C % 32 == 0, "a 32-column block lies inside one destination pixel");
    extern __shared__ __attribute__((aligned(16))) uint16_t x6s[];
    // logical tile order: (image group, position, column tile) with the column tile fastest; XCD x owns a contiguous run
    // of tiles and its persistent workgroups (slots_per_xcd of them: two per CU) walk that run with stride slots_per_xcd,
    // so at any moment the workgroups of an XCD work on ~64 consecutive positions of one image group: the TAPS^2-fold
    // re-reads of dz pixels by neighbouring positions hit that XCD's L2
    const int xcd = blockIdx.x & 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // dither (gemmx6.hip.h): the rows of a tile are IMAGES; every other group of 8 is staged negated and un-negated in the epilogue
    const bool sg_odd = !PA && (dither & 1) && (__builtin_amdgcn_readfirstlane(tid >> 6) & 1);
    const uint32_t sg_k = sg_odd ? 0x80008000u : 0x8000u;
    const float sg_s = sg_odd ? -1.f : 1.f;
    // dither & 2: which row a thread stages is permuted inside each group of 8 (b -> (b & 1) * 4 + (b >> 1)), so that the 16 lanes of a
    // ds_write_b64 group (8 lanes of a ds_write_b128 group) hold rows r and r + 4 instead of r and r + 1: with the 80-byte row pitch
    // their bank windows are then disjoint (r + 1 overlaps r on 4 of 32 banks: every staging store took two LDS cycles per group)
    const bool rowperm = (dither & 2) != 0;
    auto stage_row = [&](int r) { return rowperm ? ((r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3)) : r; };
    const int arow = stage_row(tid >> 3);
    if (dbg & 1) { hmask = nullptr; mbits = nullptr; }     // timing experiments (MRL_DGX6_DBG): 1 = no mask loads,
    // operand registers of the next k tile live ACROSS tiles (round 5): the last k step of a tile requests the first k tile of the
    // workgroup's next tile instead of re-reading its own (each tile used to start by waiting out a full memory latency)
    float4 ra0[PA ? 1 : NA];
    u32x4v rp0[PA ? 3 * NAP : 1];
    u32x4v rb0[3 * NQ];
    int t_pref = -1;                                       // >= 0: that k tile of the upcoming tile is already in flight
    for (long slot = blockIdx.x >> 3; slot < tiles_per_xcd; slot += slots_per_xcd) {      // 2 = no stores, 4 = no main loop
    const long lt = (long)xcd * tiles_per_xcd + slot;
    if (lt >= total_tiles) break;
    const int nt_i = (int)(lt % NTN);
    const long rest = lt / NTN;
    const int pos = (int)(rest % G::NPOS);
    const int bt = (int)(rest / G::NPOS);
    const int yy = pos / G::WX, xx = pos - yy * G::WX;
    const int b0 = bt * BM, n0 = nt_i * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // A rows = images b0 + p*32 + tid/8 at dz pixel (yy, xx); a tap moves the pointer by a row-independent offset
    const float* ap[NA];
    const uint16_t* app[NAP];          // PA: images b0 + p*64 + tid/4, 8 bf16 at k = (tid&3)*8, plane 0
    if constexpr (PA) {
#pragma unroll
        for (int p = 0; p < NAP; ++p) {
            const int b = min(b0 + p * 64 + (tid >> 2), B - 1);
            app[p] = dzp + (kPlanesInterleaved ? 3 : 1) * (((long)(b * OH + yy) * OW + xx) * NF) + (tid & 3) * 8;
        }
    } else {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
        const int b = min(b0 + p * 32 + arow, B - 1);
        ap[p] = dz + ((long)(b * OH + yy) * OW + xx) * NF + (tid & 7) * 4;
    }
    }
    const uint16_t* bp[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = q * 256 + tid;
        bp[q] = Bp + (long)min(n0 + stage_row(c >> 2), G::N - 1) * G::K + (c & 3) * 8;
    }
    constexpr long bplane = (long)G::N * G::K;
    // k tiles whose tap reads inside the dz map at this position (uniform over the workgroup)
    auto kvalid = [&](int t) {
        const int tap = t / G::KT_PER_TAP;
        const int a = tap / TAPS, b2 = tap - a * TAPS;
        return (unsigned)(yy - a) < (unsigned)OH && (unsigned)(xx - b2) < (unsigned)OW;
    };
    auto next_valid = [&](int t) {
        while (t < G::NKT && !kvalid(t)) ++t;
        return t;
    };
    auto fetch = [&](float4 (&ra)[PA ? 1 : NA], u32x4v (&rp)[PA ? 3 * NAP : 1], u32x4v (&rb)[3 * NQ], int tt) {      // tt: a VALID k tile
        const int tap = tt / G::KT_PER_TAP, kin = (tt - tap * G::KT_PER_TAP) * X6_BK;
        const int a = tap / TAPS, b2 = tap - a * TAPS;
        const long ko = (long)kin - (long)(a * OW + b2) * NF;
        if constexpr (PA) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int p = 0; p < NAP; ++p) rp[pl * NAP + p] = *reinterpret_cast<const u32x4v*>(app[p] + plane_off(ko, pl, dz_ps));
        } else {
#pragma unroll
        for (int p = 0; p < NA; ++p) ra[p] = *reinterpret_cast<const float4*>(ap[p] + ko);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < NQ; ++q) rb[pl * NQ + q] = *reinterpret_cast<const u32x4v*>(bp[q] + pl * bplane + tt * X6_BK);
    };
    auto swrite = [&](const float4 (&ra)[PA ? 1 : NA], const u32x4v (&rp)[PA ? 3 * NAP : 1], const u32x4v (&rb)[3 * NQ], uint16_t* As) {
        uint16_t* Bs = As + 3 * BM * X6_LDK;
        if constexpr (PA) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int p = 0; p < NAP; ++p)
                    *reinterpret_cast<u32x4v*>(As + (pl * BM + p * 64 + (tid >> 2)) * X6_LDK + (tid & 3) * 8) = rp[pl * NAP + p];
        } else {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            uint32_t a0x, a1x, a2x, a0y, a1y, a2y;
            split2_bf16x3_sg(ra[p].x, ra[p].y, sg_k, sg_s, a0x, a1x, a2x);
            split2_bf16x3_sg(ra[p].z, ra[p].w, sg_k, sg_s, a0y, a1y, a2y);
            uint16_t* d = As + (p * 32 + arow) * X6_LDK + (tid & 7) * 4;
            *reinterpret_cast<uint2*>(d) = make_uint2(a0x, a0y);
            *reinterpret_cast<uint2*>(d + BM * X6_LDK) = make_uint2(a1x, a1y);
            *reinterpret_cast<uint2*>(d + 2 * BM * X6_LDK) = make_uint2(a2x, a2y);
        }
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = q * 256 + tid;
                *reinterpret_cast<u32x4v*>(Bs + (pl * BN + stage_row(c >> 2)) * X6_LDK + (c & 3) * 8) = rb[pl * NQ + q];
            }
    };
    auto mma = [&](const bf16x8& a, const bf16x8& b, const f32x16& c) {
        if constexpr (TR) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    };
    auto mfma_block = [&](const uint16_t* As) {
        const uint16_t* Bs = As + 3 * BM * X6_LDK;
#pragma unroll
        for (int kb = 0; kb < X6_BK / 16; ++kb) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fa[a][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BM + (wm * 2 + a) * 32 + i) * X6_LDK + kb * 16 + 8 * h);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fb[b][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN + (wn * 2 + b) * 32 + i) * X6_LDK + kb * 16 + 8 * h);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {       // small terms first
                    if (X8 && kCross21) {
                        acc[a][b] = mma(fa[a][2], fb[b][1], acc[a][b]);
                        acc[a][b] = mma(fa[a][1], fb[b][2], acc[a][b]);
                    }
                    acc[a][b] = mma(fa[a][2], fb[b][0], acc[a][b]);
                    acc[a][b] = mma(fa[a][1], fb[b][1], acc[a][b]);
                    acc[a][b] = mma(fa[a][0], fb[b][2], acc[a][b]);
                    acc[a][b] = mma(fa[a][1], fb[b][0], acc[a][b]);
                    acc[a][b] = mma(fa[a][0], fb[b][1], acc[a][b]);
                    acc[a][b] = mma(fa[a][0], fb[b][0], acc[a][b]);
                }
        }
    };
    uint16_t* L0 = x6s;
    // TR + bit masks: the four mask words of the epilogue are requested during the tile's last k step (they used to be the first thing
    // the epilogue waited for)
    uint32_t mw00 = 0, mw01 = 0, mw10 = 0, mw11 = 0;
    bool mw_done = false;
    int t = (dbg & 4) ? G::NKT : next_valid(0);
    if (t < G::NKT && t_pref != t) fetch(ra0, rp0, rb0, t);
    t_pref = -1;
    while (t < G::NKT) {
        __syncthreads();                       // previous tile's fragment reads are done
        swrite(ra0, rp0, rb0, L0);
        __syncthreads();
        const int tn = next_valid(t + 1);
        int tf = tn < G::NKT ? tn : t;         // what this step's loads fetch (past the end of the last tile: a re-read, never consumed)
        // (measured: conv2's 128 x 128 tiles 4.98 -> 4.65 ms; conv3's 256 x 64 tiles, 18 k steps per tile, 3.04 -> 3.13 with 6 spilled registers: off there)
        if constexpr (!PA && NTN == 1 && WM == 2) {
            if (tn >= G::NKT) {                // last k step of the tile: its loads fetch the next tile's first valid k tile
                const long slot2 = slot + slots_per_xcd, lt2 = (long)xcd * tiles_per_xcd + slot2;
                if (slot2 < tiles_per_xcd && lt2 < total_tiles) {
                    const int pos2 = (int)(lt2 % G::NPOS), bt2 = (int)(lt2 / G::NPOS);
                    const int yy2 = pos2 / G::WX, xx2 = pos2 - yy2 * G::WX;
                    int t2 = 0;
                    while (t2 < G::NKT) {
                        const int tap = t2 / G::KT_PER_TAP, a = tap / TAPS, b2 = tap - a * TAPS;
                        if ((unsigned)(yy2 - a) < (unsigned)OH && (unsigned)(xx2 - b2) < (unsigned)OW) break;
                        ++t2;
                    }
                    if (t2 < G::NKT) {
                        // (the current tile's row pointers are dead from here on: every k tile of it has been requested)
#pragma unroll
                        for (int p = 0; p < NA; ++p) {
                            const int b = min(bt2 * BM + p * 32 + arow, B - 1);
                            ap[p] = dz + ((long)(b * OH + yy2) * OW + xx2) * NF + (tid & 7) * 4;
                        }
                        tf = t2;
                        t_pref = t2;
                    }
                }
            }
        }
        if constexpr (TR) {
            if (tn >= G::NKT && mbits && !(dbg & 2)) {
                { DGX6_EPI_ADDR(0, 0, o_, k_, c_) mw00 = k_ ? mbits[o_ >> 5] : 0u; (void)c_; }
                { DGX6_EPI_ADDR(0, 1, o_, k_, c_) mw01 = k_ ? mbits[o_ >> 5] : 0u; (void)c_; }
                { DGX6_EPI_ADDR(1, 0, o_, k_, c_) mw10 = k_ ? mbits[o_ >> 5] : 0u; (void)c_; }
                { DGX6_EPI_ADDR(1, 1, o_, k_, c_) mw11 = k_ ? mbits[o_ >> 5] : 0u; (void)c_; }
                mw_done = true;
            }
        }
        if constexpr (IL) {
            constexpr int NL = (PA ? 3 * NAP : NA) + 3 * NQ;
            static_assert(2 * NL <= 32, "two MFMAs per load inside the first half of the block");
            __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_sched_barrier(0);
            fetch(ra0, rp0, rb0, tf);
            mfma_block(L0);
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            if constexpr (WM == 2 && WN == 2) {     // 128 x 128 tiles: the second k half's fragment reads 8 MFMAs early (gemmx6.hip.h)
                static_assert(2 * NL <= 24, "room for the early fragment reads");
                __builtin_amdgcn_sched_group_barrier(0x008, 24 - 2 * NL, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 32 - 2 * NL, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            t = tn;
            continue;
        }
        fetch(ra0, rp0, rb0, tf);                    // next valid tile (or the next TILE's first one) in flight during the MFMA block
        __builtin_amdgcn_sched_barrier(0);
        if (prio) __builtin_amdgcn_s_setprio(1);     // the MFMA stream outranks the co-resident workgroup's staging VALU
        mfma_block(L0);
        if (prio) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        t = tn;
    }
    // epilogue: column (class, c) of row (image, position) -> its destination pixel, masked by act'(h) of the layer below
    if constexpr (TR) {
        // transposed accumulators: lane (i, h) owns image b0 + .. + i and channels 8g + 4h + j of one destination pixel
        const float sgn = (!PA && (dither & 1) && (i & 8)) ? -1.f : 1.f;
        if (mbits) {
            // bit-mask path: one mask word per block -- in flight since the last k step (mw_done), or requested here; no fp32 mask values
            const TrMaskRelu ef{dx, 0, nullptr, mbits, dxp, dx_ps};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    DGX6_EPI_ADDR(a, b, o_, k_, c_)
                    (void)c_;
                    const bool kk_ = k_ && (!(dbg & 2) || acc[a][b][0] == 12345.678f);   // dbg 2: no epilogue memory traffic
                    TrAux x;
                    x.w = mw_done ? (a ? (b ? mw11 : mw10) : (b ? mw01 : mw00)) : (kk_ ? mbits[o_ >> 5] : 0u);
                    tr_block_epilogue(ef, acc[a][b], x, kk_ ? o_ : 0L, h, kk_, sgn);
                }
        } else {
            const TrMaskRelu ef{dx, 0, hmask, mbits, dxp, dx_ps};
            TrAux aux[2][2];
            long oo[2][2];
            bool vv[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    DGX6_EPI_ADDR(a, b, o_, k_, c_)
                    vv[a][b] = k_ && (!(dbg & 2) || acc[a][b][0] == 12345.678f);   // dbg 2: no epilogue memory traffic
                    oo[a][b] = vv[a][b] ? o_ : 0L;
                    aux[a][b] = ef.load_aux(oo[a][b], c_, h, vv[a][b]);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) tr_block_epilogue(ef, acc[a][b], aux[a][b], oo[a][b], h, vv[a][b], sgn);
        }
    } else {
    // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (!PA && (dither & 1)) {                   // registers with r & 4 hold the images 8..15, 24..31 that were staged negated
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r & 4) acc[a][b][r] = -acc[a][b][r];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + (wn * 2 + b) * 32 + i;
            const int cls = col / C, c = col - cls * C;
            const int py = cls / S, px = cls - py * S;
            const int iy = yy * S + py, ix = xx * S + px;
            const bool colok = col < G::N && iy < H && ix < W;
            const long pix = colok ? ((long)iy * W + ix) * C + c : 0;
            long o[16];
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bimg = b0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                o[r] = (colok && bimg < B) ? (long)bimg * (H * W * C) + pix : -1;
            }
            if (mbits) {
                // ReLU bit mask of the layer below (written by its forward epilogue): the 32 lanes of a half-wave hold 32
                // consecutive channels of ONE pixel = one mask word, read as a broadcast; lane i tests bit i
                uint32_t wv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) wv[r] = mbits[(o[r] < 0 ? 0 : o[r] - i) >> 5];
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = ((wv[r] >> i) & 1u) ? 1.f : 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = hmask ? act_bwd_from_out(hmask[o[r] < 0 ? 0 : o[r]], act) : 1.f;   // all loads first
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (o[r] >= 0 && (!(dbg & 2) || acc[a][b][r] == 12345.678f)) dx[o[r]] = acc[a][b][r] * x[r];
        }
    }
    __syncthreads();          // the next tile's first LDS write must not overtake this tile's last fragment reads
    }
}

// ---- round 6: the product form with its output stores OFF the critical path ------------------------------------------------
// What the ISA of dgrad_x6_kernel showed (profiles/README.md, round 6): vmcnt counts loads and stores in ONE in-order counter, and the
// compiler's wait insertion merges states at loop headers conservatively.  The tile loop's header was reached from the kernel entry
// (operands of the first k tile in flight) and from the back edge (the same loads, then 16 epilogue stores per lane) -- so the first
// use of a prefetched operand waited with the entry path's count, i.e. for the stores as well; the runtime choice between the fp32
// activation and the bit mask of the layer below added copies of never-loaded registers behind an `s_waitcnt vmcnt(0)` right at the
// loop top.  Every tile therefore began by waiting out the write round trip of the previous tile's 64 KB (phase omission had said
// "stores: 1.4 of 4.6 ms" without saying why).  This kernel is the product configuration only (transposed epilogue, bit-mask act',
// loads between the MFMAs, no plane tensors, whole tiles: B % BM == 0, every class pixel inside the input) with the control flow
// shaped so that the wait counts are exact:
//   * no runtime variants -> no merge points with differing memory events; epilogue stores are unconditional;
//   * the state at the tile loop's header is the same on both edges: the kernel entry issues the first operand loads and then a
//     DUMMY epilogue (zeros to the first tile's own destinations, overwritten by its real epilogue -- same wave, same addresses,
//     in order), so "operand loads, then 16 stores" is what the header sees from either side and the first k step waits with
//     vmcnt(16 + n) instead of vmcnt(n);
//   * the first k step of a tile is peeled: its waits are the tile-header counts, the steady-state steps keep their own;
//   * the last k step ALWAYS issues ten operand loads (the next tile's first k tile, or a harmless re-read at the very end).
// The stores of tile T now drain while the first k step of tile T+1 multiplies; the second step's wait is the first that needs them.
// Same loads, same staging, same MFMA order, same epilogue arithmetic: bit-identical to dgrad_x6_kernel<.., TR, IL>.
template <int H, int W, int C, int RF, int S, int NF, int WM, int WN>
__global__ __launch_bounds__(256, 2) void dgrad_x6p_kernel(const float* __restrict__ dz, const uint16_t* __restrict__ Bp,
                                                        const uint32_t* __restrict__ mbits, float* __restrict__ dx, int B,
                                                        long tiles_per_xcd, long total_tiles, int slots_per_xcd, int dither) {
    using G = DgX6Geom<H, W, C, RF, S, NF>;
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = WM * 64, BN = WN * 64;
    static_assert(G::N == BN, "one column tile: the weight-plane pointers are tile-independent");
    static_assert(C % 32 == 0 && H % S == 0 && W % S == 0, "every (class, position) is a pixel of the input: unconditional stores");
    static_assert(G::KT_PER_TAP >= 2, "a tile has at least two k steps: the peeled first step is never the last");
    constexpr int NA = BM / 32, NQ = BN / 64;
    constexpr int OH = G::OH, OW = G::OW, TAPS = G::TAPS;
    constexpr bool NTS = S > 1;                          // non-temporal output stores: conv2 (see the epilogue)
    extern __shared__ __attribute__((aligned(16))) uint16_t x6s[];
    const int xcd = blockIdx.x & 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const bool sg_odd = (dither & 1) && (__builtin_amdgcn_readfirstlane(tid >> 6) & 1);
    const uint32_t sg_k = sg_odd ? 0x80008000u : 0x8000u;
    const float sg_s = sg_odd ? -1.f : 1.f;
    const bool rowperm = (dither & 2) != 0;
    auto stage_row = [&](int r) { return rowperm ? ((r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3)) : r; };
    const int arow = stage_row(tid >> 3);
    const float sgn = ((dither & 1) && (i & 8)) ? -1.f : 1.f;
    long slot = blockIdx.x >> 3;
    if (slot >= tiles_per_xcd || (long)xcd * tiles_per_xcd + slot >= total_tiles) return;

    auto first_valid = [&](int yy, int xx, int t) {      // first k tile >= t whose tap reads inside the dz map at (yy, xx)
        while (t < G::NKT) {
            const int tap = t / G::KT_PER_TAP, a = tap / TAPS, b2 = tap - a * TAPS;
            if ((unsigned)(yy - a) < (unsigned)OH && (unsigned)(xx - b2) < (unsigned)OW) break;
            ++t;
        }
        return t;
    };
    const float* ap[NA];
    auto set_rows = [&](int b0, int yy, int xx) {
#pragma unroll
        for (int p = 0; p < NA; ++p) ap[p] = dz + ((long)((b0 + p * 32 + arow) * OH + yy) * OW + xx) * NF + (tid & 7) * 4;
    };
    const uint16_t* bp[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = q * 256 + tid;
        bp[q] = Bp + (long)stage_row(c >> 2) * X6_BK + (c & 3) * 8;       // k-tile-major planes (dgx6_split_planes_kernel, kperm = 2)
    }
    constexpr long bplane = (long)G::N * G::K;
    float4 ra0[NA];
    u32x4v rb0[3 * NQ];
    auto fetch = [&](int tt) {                           // tt: a VALID k tile of the tile `ap` points at
        const int tap = tt / G::KT_PER_TAP, kin = (tt - tap * G::KT_PER_TAP) * X6_BK;
        const int a = tap / TAPS, b2 = tap - a * TAPS;
        const long ko = (long)kin - (long)(a * OW + b2) * NF;
#pragma unroll
        for (int p = 0; p < NA; ++p) ra0[p] = *reinterpret_cast<const float4*>(ap[p] + ko);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < NQ; ++q) rb0[pl * NQ + q] = *reinterpret_cast<const u32x4v*>(bp[q] + pl * bplane + (long)tt * (G::N * X6_BK));
    };
    uint16_t* const As = x6s;
    uint16_t* const Bs = As + 3 * BM * X6_LDK;
    auto swrite = [&]() {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            uint32_t a0x, a1x, a2x, a0y, a1y, a2y;
            split2_bf16x3_sg(ra0[p].x, ra0[p].y, sg_k, sg_s, a0x, a1x, a2x);
            split2_bf16x3_sg(ra0[p].z, ra0[p].w, sg_k, sg_s, a0y, a1y, a2y);
            uint16_t* d = As + (p * 32 + arow) * X6_LDK + (tid & 7) * 4;
            *reinterpret_cast<uint2*>(d) = make_uint2(a0x, a0y);
            *reinterpret_cast<uint2*>(d + BM * X6_LDK) = make_uint2(a1x, a1y);
            *reinterpret_cast<uint2*>(d + 2 * BM * X6_LDK) = make_uint2(a2x, a2y);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = q * 256 + tid;
                *reinterpret_cast<u32x4v*>(Bs + (pl * BN + stage_row(c >> 2)) * X6_LDK + (c & 3) * 8) = rb0[pl * NQ + q];
            }
    };
    f32x16 acc[2][2];
    auto mfma_block = [&]() {                            // transposed: D^T = B A^T
#pragma unroll
        for (int kb = 0; kb < X6_BK / 16; ++kb) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fa[a][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BM + (wm * 2 + a) * 32 + i) * X6_LDK + kb * 16 + 8 * h);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fb[b][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN + (wn * 2 + b) * 32 + i) * X6_LDK + kb * 16 + 8 * h);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {            // small terms first
                    if (kCross21) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][1], fa[a][2], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][2], fa[a][1], acc[a][b], 0, 0, 0);
                    }
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][0], fa[a][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][1], fa[a][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][2], fa[a][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][0], fa[a][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][1], fa[a][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][0], fa[a][0], acc[a][b], 0, 0, 0);
                }
        }
    };
    // one k step: loads of k tile `tf` issued between the MFMAs of the staged one (schedule of dgrad_x6_kernel<.., IL>)
    auto step = [&](int tf) {
        constexpr int NL = NA + 3 * NQ;
        static_assert(2 * NL <= 24, "two MFMAs per load inside the first half of the block");
        __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(tf);
        mfma_block();
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if constexpr (WM == 2 && WN == 2) {              // 128 x 128 tiles: the second k half's fragment reads 8 MFMAs early
            __builtin_amdgcn_sched_group_barrier(0x008, 24 - 2 * NL, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 32 - 2 * NL, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
    };
    // element offset of block (a, b) of this lane's image at class position (yy, xx) of tile b0
    auto epi_off = [&](int a, int b, int b0, int yy, int xx) {
        const int cb = (wn * 2 + b) * 32;
        const int cls = cb / C, c0 = cb - cls * C;
        const int py = cls / S, px = cls - py * S;
        const int iy = yy * S + py, ix = xx * S + px;
        const int bimg = b0 + (wm * 2 + a) * 32 + i;
        return (long)bimg * (H * W * C) + ((long)iy * W + ix) * C + c0;
    };

    long lt = (long)xcd * tiles_per_xcd + slot;
    int pos = (int)(lt % G::NPOS), b0 = (int)(lt / G::NPOS) * BM;
    int yy = pos / G::WX, xx = pos - yy * G::WX;
    set_rows(b0, yy, xx);
    int t = first_valid(yy, xx, 0);
    fetch(t);
    {   // dummy epilogue (see the header comment): the memory events of a real one, zeros to this tile's own destinations
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float* d = dx + epi_off(a, b, b0, yy, xx) + 4 * h + 8 * (i & 3) - (long)(i & 3) * (H * W * C);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    if constexpr (NTS) __builtin_nontemporal_store(z4, reinterpret_cast<f32x4v*>(d + (long)s2 * (H * W * C)));
                    else *reinterpret_cast<f32x4v*>(d + (long)s2 * (H * W * C)) = z4;
                }
            }
    }
    for (;;) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        // ---- first k step (peeled; never the last)
        __syncthreads();                       // the previous tile's fragment reads are done
        swrite();
        __syncthreads();
        t = first_valid(yy, xx, t + 1);
        step(t);
        // ---- remaining k steps; the last one requests the mask words of this tile's epilogue and the next tile's first operands
        uint32_t mw[2][2];
        bool more;
        int yy2 = yy, xx2 = xx, b02 = b0;
        for (;;) {
            __syncthreads();
            swrite();
            __syncthreads();
            const int tn = first_valid(yy, xx, t + 1);
            int tf = tn;
            const bool last = tn >= G::NKT;
            if (last) {
                const long slot2 = slot + slots_per_xcd, lt2 = (long)xcd * tiles_per_xcd + slot2;
                more = slot2 < tiles_per_xcd && lt2 < total_tiles;
                tf = t;                        // nothing follows: a re-read of this k tile, never consumed
                if (more) {
                    const int pos2 = (int)(lt2 % G::NPOS);
                    b02 = (int)(lt2 / G::NPOS) * BM;
                    yy2 = pos2 / G::WX; xx2 = pos2 - yy2 * G::WX;
                    set_rows(b02, yy2, xx2);   // (this tile's row pointers are dead: every k tile of it has been requested)
                    tf = first_valid(yy2, xx2, 0);
                    slot = slot2;
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) mw[a][b] = mbits[epi_off(a, b, b0, yy, xx) >> 5];
            }
            step(tf);
            if (last) { t = tf; break; }
            t = tn;
        }
        // ---- epilogue: lane (i, h) owns image i of its block and the channels 8g + 4h + j of one destination pixel
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                // whole-line stores (planes.hip.h, quad_transpose4): after the exchange this lane holds chunk 2 m + h of the images
                // i0 + s of its quad (m = i & 3, i0 = i - m); instruction s writes image i0 + s
                float* d = dx + epi_off(a, b, b0, yy, xx) + 4 * h + 8 * (i & 3) - (long)(i & 3) * (H * W * C);
                const uint32_t w = mw[a][b];
                float4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cin = 8 * g + 4 * h;
                    v[g].x = acc[a][b][4 * g] * (((w >> cin) & 1u) ? sgn : 0.f);
                    v[g].y = acc[a][b][4 * g + 1] * (((w >> (cin + 1)) & 1u) ? sgn : 0.f);
                    v[g].z = acc[a][b][4 * g + 2] * (((w >> (cin + 2)) & 1u) ? sgn : 0.f);
                    v[g].w = acc[a][b][4 * g + 3] * (((w >> (cin + 3)) & 1u) ? sgn : 0.f);
                }
                quad_transpose4(v, lane);
                // non-temporal: with whole lines per instruction the output stream no longer needs L2 to merge its pieces, and keeping
                // it out of the way leaves dz re-reads in L2 (c2.dgrad: fetch traffic 5.8 -> 4.3 GB per launch, 4.21 -> 4.13 ms; with the
                // 32-byte pieces of round 5 the same hint cost 40 %); conv3's 2.7 GB output: 2.78 -> 2.81 ms with it, plain stores there
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    typedef float f32x4v __attribute__((ext_vector_type(4)));
                    const f32x4v vv = {v[s2].x, v[s2].y, v[s2].z, v[s2].w};
                    if constexpr (NTS) __builtin_nontemporal_store(vv, reinterpret_cast<f32x4v*>(d + (long)s2 * (H * W * C)));
                    else *reinterpret_cast<f32x4v*>(d + (long)s2 * (H * W * C)) = vv;
                }
            }
        if (!more) break;
        yy = yy2; xx = xx2; b0 = b02;
    }
}

template <int H, int W, int C, int RF, int S, int NF>
inline size_t dgrad_x6_plane_bytes() {
    using G = DgX6Geom<H, W, C, RF, S, NF>;
    return (size_t)3 * G::N * G::K * sizeof(uint16_t);
}

// EXP: also instantiate the plane-tensor (PA) forms and the non-interleaved transposed form (experiment builds)
template <int H, int W, int C, int RF, int S, int NF, int WM, int WN, bool EXP = false>
inline hipError_t launch_dgrad_x6(const float* dz, const float* w, const float* hmask, const uint32_t* mbits, float* dx,
                                  int act, int B, uint16_t* planes, bool x8, int num_cus, hipStream_t stream, int dbg = 0,
                                  const uint16_t* dzp = nullptr, uint16_t* dxp = nullptr, bool tr_plain = false) {
    // dzp: plane tensor of dz (stride B*OH*OW*NF); dxp: where to leave the plane tensor of dx (stride B*H*W*C); both need
    // the eight-product mode, dxp also ReLU (or nothing) below
    using G = DgX6Geom<H, W, C, RF, S, NF>;
    if (B <= 0) return hipSuccess;
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int NTN = (G::N + BN - 1) / BN;
    const bool pa = dzp && x8 && !dbg, tr = (dxp || tr_plain) && x8 && !(dbg & 1) && C % 32 == 0 && (act == ACT_RELU || (!hmask && !mbits));
    bool pipe = false;
    if constexpr (C % 32 == 0 && NTN == 1 && H % S == 0 && W % S == 0)
        pipe = tr && !pa && mbits && !dxp && !dbg && x6_il() && B % BM == 0 && dgrad_x6_pipe();
    hipLaunchKernelGGL((dgx6_split_planes_kernel<H, W, C, RF, S, NF>), dim3((G::N * G::K + 255) / 256), dim3(256), 0, stream,
                       w, planes, pipe ? 2 : pa ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int btiles = (B + BM - 1) / BM;
    const long total = (long)btiles * G::NPOS * NTN;
    const long per_xcd = (total + 7) / 8;
    if (per_xcd * 8 > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = (size_t)3 * (BM + BN) * X6_LDK * sizeof(uint16_t);
    auto launch = [&](auto kern) {
        { hipError_t er = raise_lds_limit((const void*)kern); if (er != hipSuccess) return er; }      // once per (device, kernel)
        const int slots = (int)std::min<long>(per_xcd, std::max(1, num_cus / 8) * 2L);      // two workgroups per CU
        hipLaunchKernelGGL(kern, dim3((unsigned)(slots * 8)), dim3(256), lds, stream, dz, (const uint16_t*)planes, hmask, mbits,
                           dx, act, B, btiles, per_xcd, total, slots, dbg, x6_prio(), dzp, (long)B * G::OH * G::OW * NF, dxp,
                           (long)B * H * W * C, x6_dither());
        return hipGetLastError();
    };
    if constexpr (C % 32 == 0 && NTN == 1 && H % S == 0 && W % S == 0) {
        // round 6: the product configuration with exact wait counts around the epilogue stores (dgrad_x6 = 2, the default)
        if (pipe) {
            auto kern = dgrad_x6p_kernel<H, W, C, RF, S, NF, WM, WN>;
            { hipError_t er = raise_lds_limit((const void*)kern); if (er != hipSuccess) return er; }      // once per (device, kernel)
            const int slots = (int)std::min<long>(per_xcd, std::max(1, num_cus / 8) * 2L);
            hipLaunchKernelGGL(kern, dim3((unsigned)(slots * 8)), dim3(256), lds, stream, dz, (const uint16_t*)planes, mbits, dx, B,
                               per_xcd, total, slots, x6_dither());
            return hipGetLastError();
        }
    }
    if constexpr (C % 32 == 0) {
        if constexpr (EXP) {
            if (pa && tr) return launch(dgrad_x6_kernel<H, W, C, RF, S, NF, WM, WN, true, true, true>);
            if (tr && !x6_il()) return launch(dgrad_x6_kernel<H, W, C, RF, S, NF, WM, WN, true, false, true>);
        }
        if (tr) return launch(dgrad_x6_kernel<H, W, C, RF, S, NF, WM, WN, true, false, true, true>);
    }
    if constexpr (EXP) {
        if (pa) return launch(dgrad_x6_kernel<H, W, C, RF, S, NF, WM, WN, true, true, false>);
    }
    return launch(dgrad_x6_kernel<H, W, C, RF, S, NF, WM, WN, true>);
}

}  // namespace mrl
