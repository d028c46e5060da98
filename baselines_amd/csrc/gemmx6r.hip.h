// Tiled "bf16 x 6" GEMM, split-at-the-fragment form (gfx950, round 5).  Same arithmetic, same sums in the same order as
// gemm_x6_kernel (gemmx6.hip.h) -- bit-identical outputs -- with the operand split moved out of the staging pass:
//
//   gemm_x6_kernel:   barrier | split A (15 VALU per pair) + 24 ds_write_b64 of planes | barrier | 12 ds_read_b128 + 48 MFMAs
//   gemm_x6r_kernel:  barrier | 8 ds_write_b128 of RAW fp32 A                          | barrier | per 12 / 24 MFMAs: 2 ds_read_b128
//                                                                                                   of raw A, split in registers
//
// Why (profiles/r05b_interleave_ubench.txt): the matrix pipe and the VALU share a SIMD's issue port, but a
// v_mfma_f32_32x32x16_bf16 occupies the port for one issue slot of its 32 cycles -- VALU instructions placed BETWEEN the
// MFMAs of one wave's stream mostly hide in its shadow (48 MFMAs + 320 split VALU: 7.1 ms per 4000 steps interleaved, 7.7
// when each wave runs its split as a block and only the co-resident wave overlaps it, 5.8 for the MFMAs alone), while a
// barrier-delimited staging phase of a whole workgroup serialises: 4 waves split, then 4 waves multiply.  To interleave
// the split with the MFMAs the split has to live where the MFMAs are: on the fragment path.  That is free of redundancy when
// every A row belongs to exactly ONE wave: a wave's tile is (BM / 4) rows x ALL BN columns -- 64 x 64 for the 64-filter conv
// layers (BM = 256), 32 x 128 for the fc layer (BM = 128) -- so each A element is still split once per tile.
//   * LDS holds the A tile as raw fp32 [row][32 k], row pitch 144 bytes (conflict-free 16-byte fragment reads), 36 KB
//     instead of 61 KB of planes; B planes as before ([plane][n][40 bf16]).
//   * a wave's k step is a chain of units (k block of 16, row block of 32): the raw fragment of unit u+1 is read and
//     split between the MFMAs of unit u (sched_group_barrier pins one MFMA : VPM VALU); only the first unit of a k step
//     is split in the open.
//   * x6_dither's sign alternation keys on the row: groups of 8 rows = (lane & 8) here, so the sign and
//     the rounding constant are per-lane registers instead of scalars; same 15 instructions per pair.
// Transposed-accumulator epilogue only (planes.hip.h functors: TrBiasRelu / TrMaskRelu), N % 32 == 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gemmx6.hip.h"

namespace mrl {

typedef float x6r_f4 __attribute__((ext_vector_type(4)));      // clang vector: arrays of HIP's float4 struct captured by a lambda stay in scratch
constexpr int X6R_PITCH = 36;          // floats per raw A row in LDS (32 + 4: 144 bytes, an odd multiple of 16)

template <class AF, class EF, int BN, int VPM, int PF>
__global__ __launch_bounds__(256, 2) void gemm_x6r_kernel(AF af, const uint16_t* __restrict__ Bp, EF ef, int M, int N, int K,
                                                          int mtiles, int ntiles, int pg, int dither) {
    static_assert(BN == 64 || BN == 128, "64-filter conv layers or 128-column fc tiles");
    constexpr int BM = BN == 64 ? 256 : 128;
    constexpr int RA = BM / 128;                           // 32-row blocks per wave (its rows: wave * RA * 32 ..)
    constexpr int CB = BN / 32;                            // 32-column blocks per wave: all of the tile's columns
    constexpr int NA = BM / 32;                            // float4 of A per thread and tile
    constexpr int NQ = BN / 64;                            // 16-byte chunks of B per thread, plane and tile
    constexpr int NU = 2 * RA;                             // units (k block, row block) per k step
    extern __shared__ __attribute__((aligned(16))) uint16_t x6s[];
    float* Ar = reinterpret_cast<float*>(x6s);             // [BM][X6R_PITCH] fp32
    uint16_t* Bs = x6s + BM * X6R_PITCH * 2;               // [3][BN][X6_LDK] bf16
    // tile order: gemm_x6_kernel's (XCD-aware panels, pg row panels advance through the column tiles together)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gsz = pg * ntiles, grp = slot / gsz, rem = slot - grp * gsz;
    const int nt_i = rem / pg;
    const long panel = ((long)grp * pg + (rem - nt_i * pg)) * 8 + xcd;
    if (panel >= (long)mtiles) return;
    const int mt_i = (int)panel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int m0 = mt_i * BM, n0 = nt_i * BN;

    f32x16 acc[RA][CB];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // rows 8..15 and 24..31 of every 32-row block are multiplied NEGATED (x6_dither bit 0) and un-negated in the epilogue
    const bool sg_odd = (dither & 1) && (i & 8);
    const uint32_t sg_k = sg_odd ? 0x80008000u : 0x8000u;
    const float sg_s = sg_odd ? -1.f : 1.f;
    const bool rowperm = (dither & 2) != 0;                // B staging rows (80-byte pitch): see gemm_x6_kernel
    auto stage_row = [&](int r) { return rowperm ? ((r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3)) : r; };

    // staging addresses: A rows p*32 + tid/8, 4 floats at k = (tid&7)*4;  B rows (q*256 + tid)/4, 8 bf16 at ((..)&3)*8
    const float* ap[NA];
#pragma unroll
    for (int p = 0; p < NA; ++p)
        ap[p] = static_cast<const float*>(af.p) + af.row_base(min(m0 + p * 32 + (tid >> 3), M - 1)) + (tid & 7) * 4;
    const uint16_t* bp[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = q * 256 + tid;
        bp[q] = Bp + (long)min(n0 + stage_row(c >> 2), N - 1) * K + (c & 3) * 8;
    }
    const long bplane = (long)N * K;
    const int ntile = K / X6_BK;
    // PF register sets of the next tiles' operands: with PF = 2 the loads of tile t+2 are issued while tile t is multiplied, so a
    // k step's staging never waits for memory that was requested only one MFMA block earlier
    x6r_f4 ra[PF][NA];
    u32x4v rb[PF][3 * NQ];
    // (macros, not lambdas: register arrays captured by reference lose SROA and land in scratch memory; S is a literal)
#define X6R_FETCH(t_, S)                                                                                               \
    {                                                                                                                  \
        const int k0_ = min((t_), ntile - 1) * X6_BK; /* past the end: re-read the last tile (never consumed) */       \
        const long ko_ = af.koff(k0_);                                                                                 \
        _Pragma("unroll") for (int p = 0; p < NA; ++p) ra[S][p] = *reinterpret_cast<const x6r_f4*>(ap[p] + ko_);       \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                               \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                                             \
                rb[S][pl * NQ + q] = *reinterpret_cast<const u32x4v*>(bp[q] + pl * bplane + k0_);                      \
    }
    float* aw = Ar + (tid >> 3) * X6R_PITCH + (tid & 7) * 4;
#define X6R_SWRITE(S)                                                                                                  \
    {                                                                                                                  \
        _Pragma("unroll") for (int p = 0; p < NA; ++p) *reinterpret_cast<x6r_f4*>(aw + p * 32 * X6R_PITCH) = ra[S][p]; \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                               \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                           \
                const int c_ = q * 256 + tid;                                                                          \
                *reinterpret_cast<u32x4v*>(Bs + (pl * BN + stage_row(c_ >> 2)) * X6_LDK + (c_ & 3) * 8) = rb[S][pl * NQ + q]; \
            }                                                                                                          \
    }
    // unit u of a k step: k block kb = u / RA, row block a = u % RA.  Raw fragment: lane (i, h) = row i, k = 16 kb + 8 h .. + 7
    const float* ar = Ar + (wave * RA * 32 + i) * X6R_PITCH + 8 * h;
    auto read_raw = [&](int u, x6r_f4& lo, x6r_f4& hi) {
        const float* s = ar + (u % RA) * 32 * X6R_PITCH + (u / RA) * 16;
        lo = *reinterpret_cast<const x6r_f4*>(s);
        hi = *reinterpret_cast<const x6r_f4*>(s + 4);
    };
    auto split_frag = [&](const x6r_f4& lo, const x6r_f4& hi, bf16x8 (&f)[3]) {
        u32x4v p0, p1, p2;
        uint32_t a, b, c;
        split2_bf16x3_sg(lo.x, lo.y, sg_k, sg_s, a, b, c); p0[0] = a; p1[0] = b; p2[0] = c;
        split2_bf16x3_sg(lo.z, lo.w, sg_k, sg_s, a, b, c); p0[1] = a; p1[1] = b; p2[1] = c;
        split2_bf16x3_sg(hi.x, hi.y, sg_k, sg_s, a, b, c); p0[2] = a; p1[2] = b; p2[2] = c;
        split2_bf16x3_sg(hi.z, hi.w, sg_k, sg_s, a, b, c); p0[3] = a; p1[3] = b; p2[3] = c;
        f[0] = __builtin_bit_cast(bf16x8, p0); f[1] = __builtin_bit_cast(bf16x8, p1); f[2] = __builtin_bit_cast(bf16x8, p2);
    };
    // D^T = B A^T: the first MFMA operand supplies the accumulator's register-indexed dimension (columns)
    auto mma = [&](const bf16x8& a, const bf16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0); };
    // the MFMA block of one k step on the tile in LDS
    auto mfma_block = [&]() {
        __builtin_amdgcn_s_setprio(1);
        bf16x8 fa[2][3];
        {
            x6r_f4 lo, hi;
            read_raw(0, lo, hi);
            split_frag(lo, hi, fa[0]);         // the one split of a k step that nothing hides
        }
        bf16x8 fb[CB][3];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int kb = u / RA, a = u % RA;
            if (a == 0) {
#pragma unroll
                for (int b = 0; b < CB; ++b)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fb[b][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN + b * 32 + i) * X6_LDK + kb * 16 + 8 * h);
            }
            x6r_f4 lo, hi;
            if (u + 1 < NU) read_raw(u + 1, lo, hi);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < CB; ++b) {     // small terms first, as gemm_x6_kernel
                acc[a][b] = mma(fa[u & 1][2], fb[b][0], acc[a][b]);
                acc[a][b] = mma(fa[u & 1][1], fb[b][1], acc[a][b]);
                acc[a][b] = mma(fa[u & 1][0], fb[b][2], acc[a][b]);
                acc[a][b] = mma(fa[u & 1][1], fb[b][0], acc[a][b]);
                acc[a][b] = mma(fa[u & 1][0], fb[b][1], acc[a][b]);
                acc[a][b] = mma(fa[u & 1][0], fb[b][0], acc[a][b]);
            }
            if (u + 1 < NU) {
                split_frag(lo, hi, fa[(u + 1) & 1]);
                // one MFMA : VPM VALU until the split (60 + a few moves) is placed, the rest of the unit's MFMAs behind it
#pragma unroll
                for (int q = 0; q < 6 * CB; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define X6R_STEP(t_, S)                                                                                                \
    {                                                                                                                  \
        __syncthreads(); /* previous tile's fragment reads are done */                                                 \
        X6R_SWRITE(S)                                                                                                  \
        __syncthreads();                                                                                               \
        X6R_FETCH((t_) + PF, S) /* in flight during PF MFMA blocks */                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        mfma_block();                                                                                                  \
    }
    X6R_FETCH(0, 0)
    if constexpr (PF == 2) {
        X6R_FETCH(1, 1)
        for (int t = 0; t < ntile; t += 2) {
            X6R_STEP(t, 0)
            if (t + 1 < ntile) X6R_STEP(t + 1, PF - 1)
        }
    } else {
        for (int t = 0; t < ntile; ++t) X6R_STEP(t, 0)
    }
    // transposed accumulators: lane (i, h) owns row i and columns 8g + 4h + j of each 32-column block
    TrAux aux[RA][CB];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            const int row = m0 + (wave * RA + a) * 32 + i, cb = n0 + b * 32;
            const bool valid = row < M && cb < N;
            aux[a][b] = ef.load_aux(valid ? (long)row * ef.ld + cb : 0L, cb, h, valid);
        }
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            const int row = m0 + (wave * RA + a) * 32 + i, cb = n0 + b * 32;
            const bool valid = row < M && cb < N;
            tr_block_epilogue(ef, acc[a][b], aux[a][b], valid ? (long)row * ef.ld + cb : 0L, h, valid, sg_s);
        }
}

#undef X6R_FETCH
#undef X6R_SWRITE
#undef X6R_STEP

// mrl_set_option "x6_frag" [MRL_X6_FRAG, 1]: the transposed-epilogue launches of the tiled split engine (conv2 / conv3 / fc1
// forward, fc1 data gradient) take gemm_x6r_kernel, the conv layers walk k in class-major order; 0 = gemm_x6_kernel, natural k
// order (the round-4 engine; A/B on one box).  Experiments: 2 = two register sets of loads in flight, + 4 = natural k order
// (then bit-identical to 0).
inline int& x6_frag() { static int p = getenv("MRL_X6_FRAG") ? atoi(getenv("MRL_X6_FRAG")) : 1; return p; }

template <class AF, class EF, int BN, int VPM, int PF>
inline hipError_t launch_gemm_x6r_cfg(const AF& af, const uint16_t* Bp, const EF& ef, int M, int N, int K, hipStream_t stream) {
    constexpr int BM = BN == 64 ? 256 : 128;
    const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
    const int pg = ntiles > 1 ? std::max(1, x6_pg()) : 1;
    const long blocks = (((long)mtiles + 7) / 8 + pg - 1) / pg * pg * 8 * ntiles;
    if (blocks > 0x7fffffffL) return hipErrorInvalidValue;
    const size_t lds = (size_t)BM * X6R_PITCH * sizeof(float) + (size_t)3 * BN * X6_LDK * sizeof(uint16_t);
    auto kern = gemm_x6r_kernel<AF, EF, BN, VPM, PF>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, af, Bp, ef, M, N, K, mtiles, ntiles, pg, x6_dither());
    return hipGetLastError();
}
template <class AF, class EF>
inline hipError_t launch_gemm_x6r(const AF& af, const uint16_t* Bp, const EF& ef, int M, int N, int K, hipStream_t stream) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (N % 32 != 0 || K % X6_BK != 0) return hipErrorInvalidValue;
    const bool pf2 = (x6_frag() & 3) == 2;    // two register sets of operand loads in flight (measured slower: c2.fwd 4.44 -> 4.66 ms)
    if (N <= 64) return pf2 ? launch_gemm_x6r_cfg<AF, EF, 64, 5, 2>(af, Bp, ef, M, N, K, stream)
                            : launch_gemm_x6r_cfg<AF, EF, 64, 5, 1>(af, Bp, ef, M, N, K, stream);
    return pf2 ? launch_gemm_x6r_cfg<AF, EF, 128, 3, 2>(af, Bp, ef, M, N, K, stream)
               : launch_gemm_x6r_cfg<AF, EF, 128, 3, 1>(af, Bp, ef, M, N, K, stream);
}

}  // namespace mrl
