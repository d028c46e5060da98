// Tiled "bf16 x 6" GEMM for gfx950 (MI355X): fp32-class products on the BF16 matrix pipe, fifth GEMM engine of libmrl.
// Used for the fully connected layer of NatureCNN (common/models.py:24-26 `fc(h, 'fc1', nh=512)` via a2c/utils.py:58-63),
// forward and data-gradient:   C[M][N] = A[M][K] * B[N][K]^T   with A fp32 row-major (K contiguous).
//
// Both operands are split EXACTLY into three bf16 planes, x = x0 + x1 + x2 (round-to-nearest at each level, every
// residual exact: split2_bf16x3, wres.hip.h), and the six products down to 2^-17 relative are accumulated in fp32:
//   x0w0 + (x0w1 + x1w0) + (x0w2 + x1w1 + x2w0);   dropped: x1w2 + x2w1 + x2w2 <= 2^-24 of the product -- the rounding of
// one IEEE fp32 multiply -- and 3.5e-9 of it on average, a sixth of that rounding's mean (tests/test_split_arithmetic.py).
// 6 v_mfma_f32_32x32x16_bf16 (32 cycles, 16 k) replace 8 v_mfma_f32_32x32x2_f32 (64 cycles): 2.7x less matrix time.  Not the
// bitwise fmaf chain of the fp32 MFMA engines (gemm.hip.h stays the reference path, MRL_F32_BF16X6=0 selects it; builds
// with -DMRL_PRODUCTS8 keep x1w2 and x2w1 too: < 2^-33 per product, 8/6 of the matrix time); parity tests hold unchanged.
//
//   * B (the weight matrix, 1.6 M elements) is split and laid out [plane][n][k] ONCE per call by split_planes_kernel
//     -- its staging is then a plain 16-byte copy;
//   * A (activations / incoming gradients) is split while it is staged: a tile element is split once and used by the
//     4 column blocks of the 128-wide tile, so the VALU work is ~1/4 of a per-fragment split;
//   * 128 x 128 x 32 tiles, 4 waves x (64 x 64), LDS rows padded to 40 bf16 (conflict-free ds_read_b128 fragments),
//     60 KB LDS and ~180 VGPRs per workgroup: two workgroups per CU cover each other's barriers;
//   * XCD-aware tile order and three-phase epilogues as in gemm.hip.h (same epilogue functors).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gemm.hip.h"
#include "wres.hip.h"      // bf16x8, U32x4, split2_bf16x3
#include "planes.hip.h"    // pre-split activation planes: perm32, transposed-accumulator epilogue

namespace mrl {

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
constexpr int X6_BM = 128, X6_BN = 128, X6_BK = 32, X6_LDK = 40;

// Class-major k order of a conv forward GEMM (option x6_frag; rf % stride == 0, C % 32 == 0).  k step s (32 channels of one
// filter tap) is enumerated as (stride-parity class (py, px), 32-channel chunk kc, tap (a, c) of the class): ky = py + stride a,
// kx = px + stride c.  The taps of one class read the SAME input pixels shifted by whole pixels of the class grid, so the
// rf^2 / stride^2 k steps of a class re-read one ~43-58 KB footprint per tile back to back -- inside the XCD's 4 MB L2 for its 64
// concurrent tiles -- instead of coming back to a pixel 8 k steps (and ~16 MB of other tiles' traffic) later: counters before
// (profiles/r04z_pmc_hbm.json): conv2 forward fetched 16.8 GB for 6.7 GB of input, conv3 9.7 for 2.7, both at ~4 TB/s = HBM-bound
// on their re-reads.  A dot product does not care in which order k runs; the weight planes are laid out in the same order.
__host__ __device__ __forceinline__ void conv_kcls_decode(int s, int rf, int stride, int C, int& ky, int& kx, int& kc) {
    const int T = rf / stride, cpc = C >> 5, per = T * T * cpc;
    const int cls = s / per, r = s - cls * per;
    kc = r / (T * T);
    const int r2 = r - kc * (T * T), a = r2 / T, c = r2 - a * T;
    ky = cls / stride + stride * a;
    kx = cls % stride + stride * c;
}
// position of original k = (ky * rf + kx) * C + ch in the class-major order
__host__ __device__ __forceinline__ long conv_kcls_encode(long k, int rf, int stride, int C) {
    const int ch = (int)(k % C), t = (int)(k / C), ky = t / rf, kx = t - ky * rf;
    const int T = rf / stride, cpc = C >> 5;
    const int cls = (ky % stride) * stride + (kx % stride), a = ky / stride, c = kx / stride, kc = ch >> 5;
    return ((long)(cls * cpc + kc) * T * T + a * T + c) * 32 + (ch & 31);
}
// out[plane][n][k] (bf16 bits) from src[R][Cn] fp32:  transpose ? (n, k) = (col, row) : (n, k) = (row, col)
// kperm: k runs in the order of a plane tensor (planes.hip.h: perm32 inside each aligned block of 32)
// kcls_rf > 0: k runs in the class-major order of a conv layer with that filter size / stride / input channels (conv_kcls_encode)
// ktm (round 6): k-tile-major planes [plane][k / 32][n][k % 32] -- one k tile of ALL rows is contiguous, so the 16 rows x 64 bytes a
// wave's staging load covers are 8 whole cache lines instead of 16 half lines 2 K bytes apart (the CU's vector-memory path works per
// touched line: c2.dgrad 4.35 -> 4.13 ms with the same change, profiles/README.md round 6)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int R, int Cn, int transpose,
                                                           uint16_t* __restrict__ out, int kperm = 0, int kcls_rf = 0,
                                                           int kcls_stride = 1, int kcls_c = 32, int ktm = 0) {
    const long total = (long)R * Cn;
    const long Nn = transpose ? Cn : R, Kd = transpose ? R : Cn;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256L) {
        const long r = e / Cn, c = e - r * Cn;
        const long n = transpose ? c : r, k0 = transpose ? r : c;
        const long k = kcls_rf ? conv_kcls_encode(k0, kcls_rf, kcls_stride, kcls_c) : kperm ? kperm32(k0) : k0;
        uint16_t b0, b1, b2;
        split1_bf16x3(src[e], b0, b1, b2);
        const long eo = ktm ? (k / X6_BK) * (Nn * X6_BK) + n * X6_BK + (k % X6_BK) : n * Kd + k;
        out[0 * Nn * Kd + eo] = b0;
        out[1 * Nn * Kd + eo] = b1;
        out[2 * Nn * Kd + eo] = b2;
    }
}

// A-row descriptions: element (m, k) of the A operand lives at p[row_base(m) + koff(k)], k in chunks of X6_BK
struct X6DenseA {            // A[M][lda] row-major
    const float* p; long lda;
    __device__ __forceinline__ long row_base(int m) const { return (long)m * lda; }
    __device__ __forceinline__ long koff(int k0) const { return k0; }
};
struct X6ConvA : ConvGeom {  // conv forward: row = output pixel (b, oy, ox), k = (ky, kx, c); X6_BK divides rf*C
    int kcls = 0;            // 1: k steps in class-major order (above); the B planes must be laid out with the same order
    __device__ __forceinline__ long row_base(int m) const {
        const int ohw = OH * OW;
        int b = (int)d_ohw.div((uint32_t)m), r = m - b * ohw;
        int oy = (int)d_ow.div((uint32_t)r), ox = r - oy * OW;
        long img = srow ? (long)srow[b] : (long)b;
        return ((img * H + oy * stride) * W + ox * stride) * C;
    }
    __device__ __forceinline__ long koff(int k0) const {
        if (kcls) {
            int ky, kx, kc;
            conv_kcls_decode(k0 >> 5, rf, stride, C, ky, kx, kc);
            return ((long)ky * W + kx) * C + kc * 32;
        }
        const int ky = (int)d_rowk.div((uint32_t)k0);
        return (long)ky * W * C + (k0 - ky * rowk);
    }
};

// (WM*64) x (WN*64) x 32 tiles, WM x WN = 4 waves, each wave 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers);
// 128 x 128 (2 x 2 waves) or 256 x 64 (4 x 1, for the 64-filter conv layers).  Single LDS buffer (60 / 77 KB), two
// workgroups per CU, one register stage.  Measured alternatives (profiles/README.md): a second register stage spills
// at 2 waves per SIMD (-18 %); ONE workgroup per CU with double-buffered LDS, two register stages and the staging
// code scheduled between the MFMAs (sched_group_barrier) is 25-45 % slower -- a lone wave per SIMD stalls the matrix
// pipe at every LDS wait.
// PA: the A operand is a plane tensor (planes.hip.h; af.p = plane 0 as bf16, a_pstride elements between planes, the
//     same row / k offsets): staging is 3 x BM/64 16-byte copies per thread, no split arithmetic.
// TR: MFMA operands swapped -> a lane owns one row and 16 columns of each 32-column block; EF is a Tr* functor
//     (planes.hip.h) and the epilogue writes fp32 + planes + bit mask with 16-byte stores.  N % 32 == 0.
// IL: the global loads of the next k tile are issued BETWEEN the MFMAs of this one (sched_group_barrier pattern) instead of
//     in a phase of their own in front of the MFMA block (phase stamps: issuing 10-11 16-byte loads costs a wave 700-1400
//     cycles during which it feeds nothing to the matrix pipe).
template <class AF, class EF, int WM, int WN, bool X8, int xd = 0, bool PA = false, bool TR = false, int IL = 0>
__global__ __launch_bounds__(256) void gemm_x6_kernel(AF af, const uint16_t* __restrict__ Bp, EF ef, int M, int N, int K,
                                                      int mtiles, int ntiles, long long* dbg, int prio, long a_pstride, int pg, int dither,
                                                      int kz_tiles = 0, long zslab = 0) {
    // xd (timing experiments, builds with -DMRL_X6_EXPERIMENTS, option x6_dbg = 100 + bits): 1 = no epilogue stores, 2 = no MFMAs, 4 = no global loads in the
    // main loop, 8 = no split arithmetic (raw halves are staged), 16 = every step re-reads k tile 0 (cache hits)
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int NA = BM / 32;                            // float4 of A per thread and tile
    constexpr int NAP = BM / 64;                           // PA: 16-byte pieces of A per thread, plane and tile
    constexpr int NQ = BN / 64;                            // 16-byte chunks of B per thread, plane and tile
    extern __shared__ __attribute__((aligned(16))) uint16_t x6s[];
    // XCD-aware tile order: the column tiles of one row panel take consecutive slots of one XCD, so the A panel (the
    // operand that streams from HBM) is fetched once and hit in that XCD's L2 by the other column tiles.  The B planes
    // (9.6 MB for fc1) do not fit the 4 MB L2 and are re-read from the Infinity Cache by every row panel (PMC:
    // 10.6 GB fetched by fc1.dgrad for 1.9 GB of algorithmic reads); giving each XCD a fixed set of column tiles instead
    // keeps the planes L2-resident but makes 8 XCDs fetch every A panel: measured 27 % SLOWER (fc1.dgrad 2.9 -> 3.7 ms).
    // pg > 1 (option "x6_pg"): pg row panels of an XCD advance through the column tiles TOGETHER (slot order: column tile
    // major inside a group of pg panels), so that a B tile pulled into the XCD's L2 serves pg row panels instead of one.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gsz = pg * ntiles, grp = slot / gsz, rem = slot - grp * gsz;
    const int nt_i = rem / pg;
    const long panel = ((long)grp * pg + (rem - nt_i * pg)) * 8 + xcd;
    if (panel >= (long)mtiles) return;
    const int mt_i = (int)panel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = mt_i * BM, n0 = nt_i * BN;

    // tile-level stamps (workgroup 0): dbg[48] entry, [49] in front of the k loop, [50] behind it, [51] end of the epilogue
    auto tstamp = [&](int k) {
        if (dbg && blockIdx.x == 0 && tid == 0) dbg[48 + k] = (long long)__builtin_readcyclecounter();
    };
    tstamp(0);
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // dither (option x6_dither): every other group of 8 A rows is staged NEGATED (split2_bf16x3_sg: no extra instruction) and the
    // epilogue multiplies their results by -1 -- the matrix instruction's bias toward -inf for products far below the accumulator
    //  then changes sign every 8 rows (samples / pixels) and cancels in every later sum over rows.  A thread stages
    // rows p*32 + tid/8: groups of 8 rows = waves, so the sign and the rounding constant are wave-uniform (scalar registers).
    const bool sg_odd = !PA && (dither & 1) && (__builtin_amdgcn_readfirstlane(tid >> 6) & 1);
    const uint32_t sg_k = sg_odd ? 0x80008000u : 0x8000u;
    const float sg_s = sg_odd ? -1.f : 1.f;
    // dither & 2: which row a thread stages is permuted inside each group of 8 (b -> (b & 1) * 4 + (b >> 1)), so that the 16 lanes of a
    // ds_write_b64 group (8 lanes of a ds_write_b128 group) hold rows r and r + 4 instead of r and r + 1: with the 80-byte row pitch
    // their bank windows are then disjoint (r + 1 overlaps r on 4 of 32 banks: every staging store took two LDS cycles per group)
    const bool rowperm = (dither & 2) != 0;
    auto stage_row = [&](int r) { return rowperm ? ((r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3)) : r; };
    const int arow = stage_row(tid >> 3);
    // staging addresses: A rows p*32 + tid/8, 4 floats at k = (tid&7)*4;  B rows (q*256 + tid)/4, 8 bf16 at ((..)&3)*8
    const float* ap[NA];
    const uint16_t* app[NAP];          // PA: rows p*64 + tid/4, 8 bf16 at k = (tid&3)*8, plane 0
    if constexpr (PA) {
#pragma unroll
        for (int p = 0; p < NAP; ++p)
            app[p] = reinterpret_cast<const uint16_t*>(af.p) + (kPlanesInterleaved ? 3 : 1) * af.row_base(min(m0 + p * 64 + (tid >> 2), M - 1)) + (tid & 3) * 8;
    } else {
#pragma unroll
        for (int p = 0; p < NA; ++p)
            ap[p] = static_cast<const float*>(af.p) + af.row_base(min(m0 + p * 32 + arow, M - 1)) + (tid & 7) * 4;
    }
    const uint16_t* bp[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = q * 256 + tid;
        bp[q] = Bp + (long)min(n0 + stage_row(c >> 2), N - 1) * ((dither & 8) ? X6_BK : K) + (c & 3) * 8;      // dither & 8: k-tile-major planes
    }
    const long bplane = (long)N * K;
    const long bkt = (dither & 8) ? (long)N * X6_BK : (long)X6_BK;                                                 // distance between k tiles
    const int ntile = K / X6_BK;
    float4 ra0[PA ? 1 : NA];
    u32x4v rp0[PA ? 3 * NAP : 1];
    u32x4v rb0[3 * NQ];                // clang vector type: HIP's uint4 struct in an array is left in scratch memory by SROA
    auto fetch = [&](float4 (&ra)[PA ? 1 : NA], u32x4v (&rp)[PA ? 3 * NAP : 1], u32x4v (&rb)[3 * NQ], int t) {
        const int k0 = (xd & 16) ? 0 : min(t, ntile - 1) * X6_BK;          // past the end: re-read the last tile (never consumed)
        const long ko = af.koff(k0);
        if constexpr (PA) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int p = 0; p < NAP; ++p) rp[pl * NAP + p] = *reinterpret_cast<const u32x4v*>(app[p] + plane_off(ko, pl, a_pstride));
        } else {
#pragma unroll
            for (int p = 0; p < NA; ++p) ra[p] = *reinterpret_cast<const float4*>(ap[p] + ko);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < NQ; ++q) rb[pl * NQ + q] = *reinterpret_cast<const u32x4v*>(bp[q] + pl * bplane + (k0 / X6_BK) * bkt);
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
            if (xd & 8) {
                a0x = a1x = a2x = __float_as_uint(ra[p].x) ^ __float_as_uint(ra[p].y);
                a0y = a1y = a2y = __float_as_uint(ra[p].z) ^ __float_as_uint(ra[p].w);
            } else {
                split2_bf16x3_sg(ra[p].x, ra[p].y, sg_k, sg_s, a0x, a1x, a2x);
                split2_bf16x3_sg(ra[p].z, ra[p].w, sg_k, sg_s, a0y, a1y, a2y);
            }
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
    // TR: D^T = B A^T -- the first MFMA operand supplies the accumulator's register-indexed dimension
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
            if (xd & 2) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) acc[a][b][pl] += (float)fa[a][pl][0] + (float)fb[b][pl][0];
                continue;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {       // small terms first
                    if (X8 && kCross21) {           // -DMRL_PRODUCTS8 builds: x2 w1 and x1 w2 as well (wres.hip.h)
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
    // split-K over blockIdx.y (round 6; act-side fc launches whose tiles alone would leave half the chip idle): this workgroup
    // walks the k tiles [tz0, tz1) and writes its partial sums zslab elements behind the previous split's (kz_tiles = 0: all of K)
    const int tz0 = kz_tiles ? (int)blockIdx.y * kz_tiles : 0;
    const int tz1 = kz_tiles ? min(ntile, tz0 + kz_tiles) : ntile;
    fetch(ra0, rp0, rb0, tz0);
    // dbg != nullptr (timing experiments): wave 0 of workgroup 0 stamps the phase boundaries of its tiles 8..13
    auto stamp = [&](int t, int k) {
        if (dbg && blockIdx.x == 0 && tid == 0 && t >= 8 && t < 14) {
            dbg[(t - 8) * 8 + k] = (long long)__builtin_readcyclecounter();
            if (k == 0) dbg[(t - 8) * 8 + 6] = (long long)__builtin_amdgcn_s_memrealtime();     // 100 MHz: the shader clock follows from the two
        }
    };
    tstamp(1);
    for (int t = tz0; t < tz1; ++t) {
        stamp(t, 0);
        __syncthreads();                       // previous tile's fragment reads are done
        stamp(t, 1);
        swrite(ra0, rp0, rb0, L0);
        stamp(t, 2);
        __syncthreads();
        stamp(t, 3);
        if constexpr (IL) {
            constexpr int NL = (PA ? 3 * NAP : NA) + 3 * NQ;       // global loads per thread and k tile
            static_assert(2 * NL <= 32, "two MFMAs per load inside the first half of the block");
            stamp(t, 4);
            __builtin_amdgcn_s_setprio(1);                        // this wave's MFMA block outranks the co-resident workgroup's staging (-3 %)
            __builtin_amdgcn_sched_barrier(0);
            fetch(ra0, rp0, rb0, t + 1);
            mfma_block(L0);
            // issue order inside this region: 12 fragment reads (kb = 0), then {2 MFMAs, 1 global load} x NL, the rest of
            // kb = 0's 32 MFMAs, 12 fragment reads (kb = 1), 32 MFMAs
            constexpr int MPL = 2;                                // MFMAs per load (1: c2.fwd 5.03 -> 5.14 ms)
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPL, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            if constexpr (IL == 2) {       // kb = 1's fragment reads 8 MFMAs before the end of kb = 0 (+24 VGPRs)
                static_assert(MPL * NL <= 24, "room for the early fragment reads");
                __builtin_amdgcn_sched_group_barrier(0x008, 24 - MPL * NL, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 32 - MPL * NL, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            stamp(t, 5);
            continue;
        }
        if (!(xd & 4)) fetch(ra0, rp0, rb0, t + 1);                // next tile in flight during the MFMA block (past the end: re-reads the last)
        // fences: without them the compiler hoists the split arithmetic of swrite() up to the loads and waits for
        // them BEFORE the MFMA block (the full memory latency exposed once per tile)
        __builtin_amdgcn_sched_barrier(0);
        stamp(t, 4);
        if (prio) __builtin_amdgcn_s_setprio(1);     // experiment: the MFMA stream outranks the co-resident workgroup's staging VALU
        mfma_block(L0);
        if (prio) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp(t, 5);
    }
    tstamp(2);
    if constexpr (TR) {
        // transposed accumulators: lane (i, h) owns row i and columns 8g + 4h + j of each 32-column block
        TrAux aux[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int row = m0 + (wm * 2 + a) * 32 + i, cb = n0 + (wn * 2 + b) * 32;
                const bool valid = row < M && cb < N;
                aux[a][b] = ef.load_aux(valid ? (long)row * ef.ld + cb : 0L, cb, h, valid);
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int row = m0 + (wm * 2 + a) * 32 + i, cb = n0 + (wn * 2 + b) * 32;
                const bool valid = row < M && cb < N;
                tr_block_epilogue(ef, acc[a][b], aux[a][b], valid ? (long)row * ef.ld + cb + (long)blockIdx.y * zslab : 0L, h, valid,
                                  (!PA && (dither & 1) && (i & 8)) ? -1.f : 1.f);
            }
    } else {
    // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // ReLU bit mask (ef.mask): one 32-bit word per (row, 32-column block) = one half of a wave ballot; the 64 words of a
    // wave's 32 rows x 2 column blocks are collected into lane (row_in_block*2 + b) and stored with ONE instruction
    const int mk_row = lane >> 1, mk_b = lane & 1;
    const int mk_r = (mk_row & 3) + 4 * (mk_row >> 3), mk_h = (mk_row >> 2) & 1;
    if (!PA && (dither & 1)) {                   // rows (r&3) + 8*(r>>2) + 4h: registers with r & 4 hold the rows 8..15, 24..31 that were staged negated
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r & 4) acc[a][b][r] = -acc[a][b][r];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        uint32_t mword = 0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + (wn * 2 + b) * 32 + i;
            const int colc = min(col, N - 1);
            long o[16];
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                o[r] = (row < M && col < N) ? ef.addr(row, col, 0) : -1;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = ef.aux(o[r] < 0 ? 0 : o[r], colc);     // all loads first, unconditional
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (o[r] >= 0 && (!(xd & 1) || acc[a][b][r] == 12345.678f)) ef.put(o[r], acc[a][b][r], x[r]);
            if (ef.mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned long long bal = __ballot(o[r] >= 0 && ef.mask_bit(acc[a][b][r], x[r]));
                    if (mk_b == b && mk_r == r) mword = (uint32_t)(mk_h ? (bal >> 32) : bal);
                }
            }
        }
        if (ef.mask) {
            const int row = m0 + (wm * 2 + a) * 32 + mk_row, col = n0 + (wn * 2 + mk_b) * 32;
            if (row < M && col < N) ef.mask[ef.addr(row, col, 0) >> 5] = mword;
        }
    }
    }
    tstamp(3);
}

inline int& x6_prio() { static int p = 0; return p; }
inline int& x6_xd() { static int p = 0; return p; }        // experiment bits (mrl_set_option "x6_dbg" = 100 + bits)      // experiment knob (mrl_set_option "x6_prio")
inline bool gemm_x6_ok(const void* A, long lda, int K) {
    return K % X6_BK == 0 && lda % 4 == 0 && (uintptr_t)A % 16 == 0;
}
inline size_t gemm_x6_plane_bytes(long N, long K) { return (size_t)3 * N * K * sizeof(uint16_t); }

// transpose form through an LDS tile (round 5): out[plane][n][k] from src[k][n] -- the element-wise kernel above writes 2-byte
// pieces K * 2 bytes apart (fc1's 3136 x 512 weights: 28-35 us per call, twice per minibatch step and once per act step); here a
// workgroup reads a 32 (k) x 64 (n) tile with coalesced rows and writes 16-byte pieces of 8 consecutive k per (plane, n).
__global__ __launch_bounds__(256) void split_planes_tr_kernel(const float* __restrict__ src, int K, int N, uint16_t* __restrict__ out, int ktm) {
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    const int ntn = N / 64, tk = blockIdx.x / ntn, tn = blockIdx.x - tk * ntn;
    const int k0 = tk * 32, n0 = tn * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = (tid >> 6) + 4 * j, n = tid & 63;
        tile[k][n] = src[(long)(k0 + k) * N + n0 + n];
    }
    __syncthreads();
    const int n = tid >> 2, kc = (tid & 3) * 8;
    uint32_t p[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        split2_bf16x3(tile[kc + 2 * q][n], tile[kc + 2 * q + 1][n], p[0][q], p[1][q], p[2][q]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<u32x4v*>(out + (ktm ? (long)pl * N * K + (long)tk * (N * 32) + (long)(n0 + n) * 32 + kc : ((long)pl * N + n0 + n) * K + k0 + kc)) =
            u32x4v{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
}

inline hipError_t launch_split_planes(const float* src, int R, int Cn, bool transpose, uint16_t* out, hipStream_t stream,
                                      bool kperm = false, int kcls_rf = 0, int kcls_stride = 1, int kcls_c = 32, bool ktm = false) {
    if (ktm && (kperm || kcls_rf || (transpose ? R : Cn) % X6_BK != 0)) return hipErrorInvalidValue;
    if (transpose && !kperm && !kcls_rf && R % 32 == 0 && Cn % 64 == 0 && (uintptr_t)out % 16 == 0) {
        hipLaunchKernelGGL(split_planes_tr_kernel, dim3((unsigned)((R / 32) * (Cn / 64))), dim3(256), 0, stream, src, R, Cn, out, ktm ? 1 : 0);
        return hipGetLastError();
    }
    const long total = (long)R * Cn;
    const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, stream, src, R, Cn, transpose ? 1 : 0, out, kperm ? 1 : 0,
                       kcls_rf, kcls_stride, kcls_c, ktm ? 1 : 0);
    return hipGetLastError();
}

inline int& x6_pg() { static int p = getenv("MRL_X6_PG") ? atoi(getenv("MRL_X6_PG")) : 8; return p; }          // mrl_set_option "x6_pg": row panels per pass over the B tiles
inline int& x6_il() { static int p = getenv("MRL_X6_IL") ? atoi(getenv("MRL_X6_IL")) : 1; return p; }          // mrl_set_option "x6_il": loads interleaved with the MFMAs (TR launches)
template <class AF, class EF, int WM, int WN, bool X8, int XD = 0, bool PA = false, bool TR = false, int IL = 0>
inline hipError_t launch_gemm_x6_cfg(const AF& af, const uint16_t* Bp, const EF& ef, int M, int N, int K, long long* dbg,
                                     hipStream_t stream, long a_pstride = 0, bool ktm = false, int nz = 1, long zslab = 0) {
    constexpr int BM = WM * 64, BN = WN * 64;
    const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
    const int pg = ntiles > 1 ? std::max(1, x6_pg()) : 1;
    const long blocks = (((long)mtiles + 7) / 8 + pg - 1) / pg * pg * 8 * ntiles;
    if (blocks > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = (size_t)3 * (BM + BN) * X6_LDK * sizeof(uint16_t);
#ifdef MRL_X6_EXPERIMENTS
    if constexpr (XD == 0 && !PA && !TR) {
        if (x6_xd() != 0) switch (x6_xd()) {
        case 1: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 1>(af, Bp, ef, M, N, K, dbg, stream);
        case 2: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 2>(af, Bp, ef, M, N, K, dbg, stream);
        case 3: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 3>(af, Bp, ef, M, N, K, dbg, stream);
        case 4: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 4>(af, Bp, ef, M, N, K, dbg, stream);
        case 8: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 8>(af, Bp, ef, M, N, K, dbg, stream);
        case 7: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 7>(af, Bp, ef, M, N, K, dbg, stream);
        case 15: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 15>(af, Bp, ef, M, N, K, dbg, stream);
        case 16: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 16>(af, Bp, ef, M, N, K, dbg, stream);
        case 5: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 5>(af, Bp, ef, M, N, K, dbg, stream);
        case 17: return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, 17>(af, Bp, ef, M, N, K, dbg, stream);
        default: break;
        }
    }
#endif
    if constexpr (TR && !PA && !IL && XD == 0) {
        // 128 x 128 tiles also read the second k half's fragments 8 MFMAs early (IL = 2: fc1.fwd 2.48 -> 2.38 ms, fc1.dgrad
        // 2.88 -> 2.82); the 256 x 64 tiles of the conv layers lose 1 % with that (c2.fwd 4.82 -> 4.87)
        constexpr int ILV = (WM == 2 && WN == 2) ? 2 : 1;
#ifdef MRL_X6_EXPERIMENTS
        if (x6_il())
#endif
        return launch_gemm_x6_cfg<AF, EF, WM, WN, X8, XD, PA, TR, ILV>(af, Bp, ef, M, N, K, dbg, stream, a_pstride, ktm, nz, zslab);
    }
    auto kern = gemm_x6_kernel<AF, EF, WM, WN, X8, XD, PA, TR, IL>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    const int kz_tiles = nz > 1 ? (K / X6_BK + nz - 1) / nz : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)std::max(1, nz)), dim3(256), lds, stream, af, Bp, ef, M, N, K, mtiles, ntiles, dbg,
                       x6_prio(), a_pstride, pg, x6_dither() | (ktm ? 8 : 0), kz_tiles, zslab);
    return hipGetLastError();
}
// Pre-split operands (planes.hip.h).  PA: A is a plane tensor (af.p = plane 0, a_pstride elements between planes; Bp must
// have been laid out with kperm).  TR: EF is a Tr* functor writing fp32 + planes (+ bit mask); needs N % 32 == 0.
// Eight-product arithmetic only (the default mode).
template <bool PA, bool TR, class AF, class EF>
inline hipError_t launch_gemm_x6_planes(const AF& af, long a_pstride, const uint16_t* Bp, const EF& ef, int M, int N, int K,
                                        hipStream_t stream, long long* dbg = nullptr, bool ktm = false, int nz = 1, long zslab = 0) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (TR && N % 32 != 0) return hipErrorInvalidValue;
    if (N <= 64) return launch_gemm_x6_cfg<AF, EF, 4, 1, true, 0, PA, TR>(af, Bp, ef, M, N, K, dbg, stream, a_pstride, ktm, nz, zslab);
    return launch_gemm_x6_cfg<AF, EF, 2, 2, true, 0, PA, TR>(af, Bp, ef, M, N, K, dbg, stream, a_pstride, ktm, nz, zslab);
}
template <class AF, class EF>
inline hipError_t launch_gemm_x6(const AF& af, const uint16_t* Bp, const EF& ef, int M, int N, int K, hipStream_t stream,
                                 long long* dbg = nullptr, bool x8 = false) {
    if (M <= 0 || N <= 0) return hipSuccess;
    // 256 x 64 tiles for the 64-filter conv layers, 128 x 128 otherwise
    (void)x8;          // one arithmetic per build: kSplitProducts products per multiply (wres.hip.h)
    if (N <= 64) return launch_gemm_x6_cfg<AF, EF, 4, 1, true>(af, Bp, ef, M, N, K, dbg, stream);
    return launch_gemm_x6_cfg<AF, EF, 2, 2, true>(af, Bp, ef, M, N, K, dbg, stream);
}

}  // namespace mrl
