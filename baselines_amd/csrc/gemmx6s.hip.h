// Wave-specialised form of the tiled split-bf16 engine (gemmx6.hip.h / dgradx6.hip.h) for gfx950.
//
// What limits the plain engine (phase stamps, profiles/README.md): a tile step is ~2.6 k cycles of MFMA (8 products) but
// ~5.5 k cycles long -- the waves that issue the MFMAs are the same waves that wait for the next tile's global loads, split
// them and write them to LDS, and two co-resident workgroups drift into the same phase instead of covering each other
// (the data-gradient experiments in profiles/r02c_dgradx6_phase_experiments.txt add up as if the phases were serial).
// Here the two jobs belong to DIFFERENT waves of one 512-thread persistent workgroup per CU:
//   waves 4-7  PRODUCERS: global -> registers (two register sets: loads run two tiles ahead) -> exact 3-way bf16 split
//              -> LDS buffer (g+1)&1.  They never touch the matrix pipe.
//   waves 0-3  CONSUMERS: ds_read_b128 fragments of LDS buffer g&1 (the second half-step's fragments are read before the
//              first half-step's MFMAs are issued) -> 64 MFMAs per step -> epilogue.  They never wait for HBM.
// One SIMD hosts one wave of each kind; ONE barrier per tile step separates "buffer g written" from "buffer g-2 free";
// the step counter g runs across tiles, so while the consumers are in a tile's epilogue the producers already hold the
// next tile's first two k-tiles in registers / LDS.  Arithmetic, tile shapes, fragment layouts, the pre-split weight
// planes and the epilogue functors are those of gemmx6.hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgradx6.hip.h"

namespace mrl {

// ---- problem descriptions (what differs between the dense / conv-forward GEMM and the position-major data gradient) ----
template <class AF, class EF, int BM, int BN>
struct X6sDense {
    AF af; EF ef; int M, N, K, ntiles;
    struct Tile { int m0, n0; };
    __device__ __forceinline__ Tile tile(long lt) const {
        const int nt = (int)(lt % ntiles);
        return Tile{(int)(lt / ntiles) * BM, nt * BN};
    }
    __device__ __forceinline__ int nkt() const { return K / X6_BK; }
    __device__ __forceinline__ bool kvalid(const Tile&, int) const { return true; }
    __device__ __forceinline__ const float* a_row(const Tile& t, int r) const {
        return static_cast<const float*>(af.p) + af.row_base(min(t.m0 + r, M - 1));
    }
    __device__ __forceinline__ long a_koff(const Tile&, int kt) const { return af.koff(kt * X6_BK); }
    __device__ __forceinline__ long b_row(const Tile& t, int c) const { return (long)min(t.n0 + c, N - 1) * K; }
    __device__ __forceinline__ long b_plane() const { return (long)N * K; }
    // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    __device__ __forceinline__ void epilogue(const Tile& t, f32x16 (&acc)[2][2], int wm, int wn, int lane) const {
        const int i = lane & 31, h = lane >> 5;
        const int mk_row = lane >> 1, mk_b = lane & 1;
        const int mk_r = (mk_row & 3) + 4 * (mk_row >> 3), mk_h = (mk_row >> 2) & 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            uint32_t mword = 0;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = t.n0 + (wn * 2 + b) * 32 + i;
                const int colc = min(col, N - 1);
                long o[16];
                float x[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = t.m0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    o[r] = (row < M && col < N) ? ef.addr(row, col, 0) : -1;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = ef.aux(o[r] < 0 ? 0 : o[r], colc);     // all loads first
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (o[r] >= 0) ef.put(o[r], acc[a][b][r], x[r]);
                if (ef.mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned long long bal = __ballot(o[r] >= 0 && ef.mask_bit(acc[a][b][r], x[r]));
                        if (mk_b == b && mk_r == r) mword = (uint32_t)(mk_h ? (bal >> 32) : bal);
                    }
                }
            }
            if (ef.mask) {
                const int row = t.m0 + (wm * 2 + a) * 32 + mk_row, col = t.n0 + (wn * 2 + mk_b) * 32;
                if (row < M && col < N) ef.mask[ef.addr(row, col, 0) >> 5] = mword;
            }
        }
    }
};

template <int H, int W, int C, int RF, int S, int NF, int BM, int BN>
struct X6sDgrad {
    using G = DgX6Geom<H, W, C, RF, S, NF>;
    static constexpr int NTN = (G::N + BN - 1) / BN;
    const float* dz; const float* hmask; const uint32_t* mbits; float* dx; int act, B;
    struct Tile { int b0, n0, yy, xx; };
    __device__ __forceinline__ Tile tile(long lt) const {
        const int nt = (int)(lt % NTN);
        const long rest = lt / NTN;
        const int pos = (int)(rest % G::NPOS);
        return Tile{(int)(rest / G::NPOS) * BM, nt * BN, pos / G::WX, pos % G::WX};
    }
    __device__ __forceinline__ int nkt() const { return G::NKT; }
    __device__ __forceinline__ bool kvalid(const Tile& t, int kt) const {
        const int tap = kt / G::KT_PER_TAP;
        const int a = tap / G::TAPS, b2 = tap - a * G::TAPS;
        return (unsigned)(t.yy - a) < (unsigned)G::OH && (unsigned)(t.xx - b2) < (unsigned)G::OW;
    }
    __device__ __forceinline__ const float* a_row(const Tile& t, int r) const {
        const int b = min(t.b0 + r, B - 1);
        return dz + ((long)(b * G::OH + t.yy) * G::OW + t.xx) * NF;
    }
    __device__ __forceinline__ long a_koff(const Tile&, int kt) const {
        const int tap = kt / G::KT_PER_TAP, kin = (kt - tap * G::KT_PER_TAP) * X6_BK;
        const int a = tap / G::TAPS, b2 = tap - a * G::TAPS;
        return (long)kin - (long)(a * G::OW + b2) * NF;
    }
    __device__ __forceinline__ long b_row(const Tile& t, int c) const { return (long)min(t.n0 + c, G::N - 1) * G::K; }
    __device__ __forceinline__ long b_plane() const { return (long)G::N * G::K; }
    __device__ __forceinline__ void epilogue(const Tile& t, f32x16 (&acc)[2][2], int wm, int wn, int lane) const {
        const int i = lane & 31, h = lane >> 5;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = t.n0 + (wn * 2 + b) * 32 + i;
                const int cls = col / C, c = col - cls * C;
                const int py = cls / S, px = cls - py * S;
                const int iy = t.yy * S + py, ix = t.xx * S + px;
                const bool colok = col < G::N && iy < H && ix < W;
                const long pix = colok ? ((long)iy * W + ix) * C + c : 0;
                long o[16];
                float x[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int bimg = t.b0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    o[r] = (colok && bimg < B) ? (long)bimg * (H * W * C) + pix : -1;
                }
                if (mbits) {
                    uint32_t wv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) wv[r] = mbits[(o[r] < 0 ? 0 : o[r] - i) >> 5];
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] = ((wv[r] >> i) & 1u) ? 1.f : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r] = hmask ? act_bwd_from_out(hmask[o[r] < 0 ? 0 : o[r]], act) : 1.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (o[r] >= 0) dx[o[r]] = acc[a][b][r] * x[r];
            }
    }
};

// ---- the kernel --------------------------------------------------------------------------------------------------------
template <class P, int WM, int WN, bool X8>
__global__ __launch_bounds__(512) void x6s_kernel(P prob, const uint16_t* __restrict__ Bp, long total_tiles, long tiles_per_xcd,
                                                  int slots_per_xcd, long long* dbg) {
    static_assert(WM * WN == 4, "4 consumer waves");
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int NA = BM / 32, NQ = BN / 64;
    constexpr int BUF = 3 * (BM + BN) * X6_LDK;               // bf16 elements per LDS buffer
    extern __shared__ __attribute__((aligned(16))) uint16_t x6s_lds[];
    const int tid = threadIdx.x;
    const bool producer = tid >= 256;                          // wave-uniform: waves 4..7
    const int ptid = tid & 255, lane = tid & 63, wave = ptid >> 6;
    const int xcd = blockIdx.x & 7;
    unsigned g = 0;                                            // running tile-step counter: LDS buffer = g & 1
    // dbg != nullptr (timing experiments, MRL_X6_DBG=2): lane 0 of consumer wave 0 / producer wave 4 of workgroup 0 stamp
    // the phase boundaries of steps 16..23:  dbg[(g-16)*16 + k], k = 0..4 consumer, 8..11 producer
    auto stamp = [&](int k) {
        if (dbg && blockIdx.x == 0 && (tid == 0 || tid == 256) && g >= 16 && g < 24)
            dbg[(g - 16) * 16 + k] = (long long)__builtin_readcyclecounter();
    };

    for (long slot = blockIdx.x >> 3; slot < tiles_per_xcd; slot += slots_per_xcd) {
        const long lt = (long)xcd * tiles_per_xcd + slot;
        if (lt >= total_tiles) break;
        const typename P::Tile ts = prob.tile(lt);
        const int NKT = prob.nkt();
        auto next_valid = [&](int t) {
            while (t < NKT && !prob.kvalid(ts, t)) ++t;
            return t;
        };
        if (producer) {
            // ---------------- producers: stage k tile by k tile into LDS, loads two tiles ahead -------------------------
            const float* ap[NA];
#pragma unroll
            for (int p = 0; p < NA; ++p) ap[p] = prob.a_row(ts, p * 32 + (ptid >> 3)) + (ptid & 7) * 4;
            const uint16_t* bp[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = q * 256 + ptid;
                bp[q] = Bp + prob.b_row(ts, c >> 2) + (c & 3) * 8;
            }
            const long bplane = prob.b_plane();
            auto fetch = [&](float4 (&ra)[NA], u32x4v (&rb)[3 * NQ], int kt) {
                const long ko = prob.a_koff(ts, kt);
#pragma unroll
                for (int p = 0; p < NA; ++p) ra[p] = *reinterpret_cast<const float4*>(ap[p] + ko);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) rb[pl * NQ + q] = *reinterpret_cast<const u32x4v*>(bp[q] + pl * bplane + kt * X6_BK);
            };
            auto swrite = [&](const float4 (&ra)[NA], const u32x4v (&rb)[3 * NQ], uint16_t* As) {
                uint16_t* Bs = As + 3 * BM * X6_LDK;
#pragma unroll
                for (int p = 0; p < NA; ++p) {
                    uint32_t a0x, a1x, a2x, a0y, a1y, a2y;
                    split2_bf16x3(ra[p].x, ra[p].y, a0x, a1x, a2x);
                    split2_bf16x3(ra[p].z, ra[p].w, a0y, a1y, a2y);
                    uint16_t* d = As + (p * 32 + (ptid >> 3)) * X6_LDK + (ptid & 7) * 4;
                    *reinterpret_cast<uint2*>(d) = make_uint2(a0x, a0y);
                    *reinterpret_cast<uint2*>(d + BM * X6_LDK) = make_uint2(a1x, a1y);
                    *reinterpret_cast<uint2*>(d + 2 * BM * X6_LDK) = make_uint2(a2x, a2y);
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = q * 256 + ptid;
                        *reinterpret_cast<u32x4v*>(Bs + (pl * BN + (c >> 2)) * X6_LDK + (c & 3) * 8) = rb[pl * NQ + q];
                    }
            };
            float4 ra0[NA], ra1[NA];
            u32x4v rb0[3 * NQ], rb1[3 * NQ];
            int t0 = next_valid(0);
            int t1 = t0 < NKT ? next_valid(t0 + 1) : NKT;
            if (t0 < NKT) fetch(ra0, rb0, t0);
            if (t1 < NKT) fetch(ra1, rb1, t1);
            while (t0 < NKT) {
                // even step: register set 0
                int t2 = t1 < NKT ? next_valid(t1 + 1) : NKT;
                stamp(8);
                swrite(ra0, rb0, x6s_lds + (g & 1) * BUF);
                stamp(9);
                if (t2 < NKT) fetch(ra0, rb0, t2);
                stamp(10);
                __syncthreads();                               // buffer g written; buffer g-1 may be refilled after the next one
                stamp(11);
                ++g;
                if (t1 >= NKT) break;
                // odd step: register set 1
                int t3 = t2 < NKT ? next_valid(t2 + 1) : NKT;
                stamp(8);
                swrite(ra1, rb1, x6s_lds + (g & 1) * BUF);
                stamp(9);
                if (t3 < NKT) fetch(ra1, rb1, t3);
                stamp(10);
                __syncthreads();
                stamp(11);
                ++g;
                t0 = t2; t1 = t3;
            }
        } else {
            // ---------------- consumers: fragments + MFMA, then the epilogue ------------------------------------------------
            const int i = lane & 31, h = lane >> 5;
            const int wm = wave / WN, wn = wave % WN;
            f32x16 acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            int nsteps = 0;
            for (int t = next_valid(0); t < NKT; t = next_valid(t + 1)) ++nsteps;
            auto frags = [&](const uint16_t* As, int kb, bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3]) {
                const uint16_t* Bs = As + 3 * BM * X6_LDK;
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
            };
            auto mma = [&](const bf16x8 (&fa)[2][3], const bf16x8 (&fb)[2][3]) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {       // small terms first (accumulator-major: measured faster than taking turns)
                        if (X8) {
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][1], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][2], acc[a][b], 0, 0, 0);
                        }
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
                    }
            };
            static_assert(X6_BK == 32, "two half-steps of 16 k");
            for (int s = 0; s < nsteps; ++s) {
                stamp(0);
                __syncthreads();                               // buffer g has been written
                stamp(1);
                const uint16_t* As = x6s_lds + (g & 1) * BUF;
                bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];
                frags(As, 0, fa0, fb0);
                frags(As, 1, fa1, fb1);                        // second half-step's reads are in flight behind the first MFMAs
                stamp(2);
                mma(fa0, fb0);
                stamp(3);
                mma(fa1, fb1);
                stamp(4);
                ++g;
            }
            prob.epilogue(ts, acc, wm, wn, lane);
        }
    }
}

template <class P, int WM, int WN>
inline hipError_t launch_x6s(const P& prob, const uint16_t* Bp, long total_tiles, int num_cus, bool x8, hipStream_t stream,
                             long long* dbg = nullptr) {
    if (total_tiles <= 0) return hipSuccess;
    constexpr int BM = WM * 64, BN = WN * 64;
    const size_t lds = (size_t)2 * 3 * (BM + BN) * X6_LDK * sizeof(uint16_t);
    const long per_xcd = (total_tiles + 7) / 8;
    const int slots = (int)std::min<long>(per_xcd, std::max(1, num_cus / 8));        // one persistent workgroup per CU
    auto go = [&](auto kern) {
        { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
        hipLaunchKernelGGL(kern, dim3((unsigned)(slots * 8)), dim3(512), lds, stream, prob, Bp, total_tiles, per_xcd, slots, dbg);
        return hipGetLastError();
    };
    if (x8) return go(x6s_kernel<P, WM, WN, true>);
    return go(x6s_kernel<P, WM, WN, false>);
}

// dense / conv-forward GEMM  C[M][N] = A[M][K] * B[N][K]^T  (same interface as launch_gemm_x6)
template <class AF, class EF>
inline hipError_t launch_gemm_x6s(const AF& af, const uint16_t* Bp, const EF& ef, int M, int N, int K, int num_cus, bool x8,
                                  hipStream_t stream, long long* dbg = nullptr) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (N <= 64) {
        constexpr int BM = 256, BN = 64;
        const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
        X6sDense<AF, EF, BM, BN> p{af, ef, M, N, K, nt};
        return launch_x6s<X6sDense<AF, EF, BM, BN>, 4, 1>(p, Bp, (long)mt * nt, num_cus, x8, stream, dbg);
    }
    constexpr int BM = 128, BN = 128;
    const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
    X6sDense<AF, EF, BM, BN> p{af, ef, M, N, K, nt};
    return launch_x6s<X6sDense<AF, EF, BM, BN>, 2, 2>(p, Bp, (long)mt * nt, num_cus, x8, stream, dbg);
}

// position-major conv data gradient (same interface as launch_dgrad_x6)
template <int H, int W, int C, int RF, int S, int NF, int WM, int WN>
inline hipError_t launch_dgrad_x6s(const float* dz, const float* w, const float* hmask, const uint32_t* mbits, float* dx, int act,
                                   int B, uint16_t* planes, bool x8, int num_cus, hipStream_t stream) {
    using G = DgX6Geom<H, W, C, RF, S, NF>;
    if (B <= 0) return hipSuccess;
    constexpr int BM = WM * 64, BN = WN * 64;
    using P = X6sDgrad<H, W, C, RF, S, NF, BM, BN>;
    hipLaunchKernelGGL((dgx6_split_planes_kernel<H, W, C, RF, S, NF>), dim3((G::N * G::K + 255) / 256), dim3(256), 0, stream,
                       w, planes, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int btiles = (B + BM - 1) / BM;
    P p{dz, hmask, mbits, dx, act, B};
    return launch_x6s<P, WM, WN>(p, planes, (long)btiles * G::NPOS * P::NTN, num_cus, x8, stream);
}

}  // namespace mrl
