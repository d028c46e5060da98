// Weight gradients on the bf16 matrix pipe with fp32-class products (gfx950): dW[k][n] = sum_m A[m][k] * dZ[m][n],
// db[n] = sum_m dZ[m][n]  (the gradient of `conv` / `fc`, a2c/utils.py:37-63, taken by tf.gradients in
// ppo2/model.py:100-109), for the conv2 / conv3 / fc1 layers of NatureCNN (common/models.py:19-26).
//
// Same arithmetic as gemmx6.hip.h: both operands are split EXACTLY into three bf16 planes while they are staged, six of the
// nine partial products are accumulated in fp32 (dropped: at most 2^-24 of a product; eight in -DMRL_PRODUCTS8 builds, for
// which this engine was first written -- hence the name).
// What differs is the contraction index: it is the GEMM ROW m (sample, output pixel), which is the slow index of both
// operands in memory, while v_mfma_f32_32x32x16_bf16 wants 8 consecutive contraction elements per lane.  The transpose
// is free in the staging pass: a thread loads 16-byte pieces of 8 (or 4, or 2) consecutive rows, and the bf16 pairs it
// packs after the split are pairs of ROWS -- the same v_perm count as packing pairs of columns -- so the LDS image is
// [plane][k or n][32 m] and the fragment reads are the plain ds_read_b128 of the forward engine.
//   * LDS rows are 32 m + 8 pad bf16; the four 8-m octets of a row are XOR-swizzled with bits 4-5 of the row index, which
//     makes the transposed writes (lane = 4-row group, stride 4 rows) conflict-free as well as the fragment reads;
//   * the row loop is long (rows / slabs / 32 steps per workgroup, no tile transitions), one partial slab per workgroup
//     range, summed by reduce_slabs (model.hip) -- the split-K scheme of the fp32 weight-gradient engines;
//   * workgroups that share a row range (the output tiles of one slab) take consecutive slots of one XCD: the A rows and
//     dZ rows are fetched from HBM once and hit in that XCD's L2 by the other tiles;
//   * the bias column sums ride on the dZ staging registers (k tile 0 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemmx6.hip.h"

namespace mrl {

constexpr int WX_LDK = 40;

template <class AF, int WM, int WN, int NT>
__global__ __launch_bounds__(NT, 2) void wgrad_x8_kernel(AF af, const float* __restrict__ dz, float* __restrict__ part, long slab,
                                                         int M, int K, int N, int ktiles, int ntiles, int nslab, int steps_per_slab) {
    constexpr int BKo = WM * 64, BNo = WN * 64;
    constexpr bool A128 = (BKo == NT);                      // one 8-row task per thread, else two threads per 8 rows (4-row tasks)
    static_assert(A128 || BKo * 2 == NT, "A staging: 8-row or 4-row tasks");
    constexpr int AR = A128 ? 8 : 4;                        // rows per A task
    constexpr int BKq = BKo / 4;                            // 16-byte k chunks per A row
    constexpr int BNq = BNo / 4;
    constexpr int NBT = 4 * BNo;                            // dZ tasks (2 rows x 4 n each)
    constexpr int NQB = (NBT + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) uint16_t wxs[];
    uint16_t* As = wxs;                                     // [3][BKo][LDK]
    uint16_t* Bs = wxs + 3 * BKo * WX_LDK;                  // [3][BNo][LDK]
    const int T = ktiles * ntiles;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = slot % T, s = (slot / T) * 8 + xcd;
    if (s >= nslab) return;
    const int kt = tile / ntiles, nt = tile - kt * ntiles;
    const int k0 = kt * BKo, n0 = nt * BNo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const long m_begin = (long)s * steps_per_slab * 32;
    const long m_end = min((long)M, m_begin + (long)steps_per_slab * 32);
    const int nsteps = (int)((m_end - m_begin + 31) / 32);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- staging roles
    const int a_kc = tid % BKq, a_mg = tid / BKq;           // A: k chunk, row group (8 or 4 rows)
    const long a_ko = af.koff(min(k0 + 4 * a_kc, K - 4));
    const float* ap = static_cast<const float*>(af.p);
    int b_nc[NQB], b_ro[NQB], b_mo[NQB];
    bool b_on[NQB];
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
        const int id = q * NT + tid;
        b_on[q] = id < NBT;
        b_nc[q] = id % BNq;
        b_ro[q] = (id / BNq) & 3;
        b_mo[q] = min(id / BNo, 3);
    }
    float4 ra[AR];
    float4 rb[NQB][2];
    float bsum[NQB][4];
#pragma unroll
    for (int q = 0; q < NQB; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) bsum[q][j] = 0.f;

    auto fetch = [&](int t) {
        const long ms = m_begin + (long)min(t, nsteps - 1) * 32;     // past the end: re-read the last step (never consumed)
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const long m = min(ms + a_mg * AR + r, (long)M - 1);
            ra[r] = *reinterpret_cast<const float4*>(ap + af.row_base((int)m) + a_ko);
        }
#pragma unroll
        for (int q = 0; q < NQB; ++q)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const long m = ms + b_mo[q] * 8 + b_ro[q] * 2 + r;
                const float4 v = *reinterpret_cast<const float4*>(dz + min(m, (long)M - 1) * N + n0 + 4 * b_nc[q]);
                const float z = (m < m_end) ? 1.f : 0.f;             // rows beyond the range contribute nothing
                rb[q][r] = make_float4(v.x * z, v.y * z, v.z * z, v.w * z);
            }
    };
    auto swrite = [&]() {
        // A: for each of the 4 k of the chunk, AR consecutive rows -> AR/2 row pairs per plane
        const int sw = (a_kc >> 2) & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t p[3][AR / 2];
#pragma unroll
            for (int r = 0; r < AR; r += 2) {
                const float x0 = j == 0 ? ra[r].x : j == 1 ? ra[r].y : j == 2 ? ra[r].z : ra[r].w;
                const float x1 = j == 0 ? ra[r + 1].x : j == 1 ? ra[r + 1].y : j == 2 ? ra[r + 1].z : ra[r + 1].w;
                split2_bf16x3(x0, x1, p[0][r / 2], p[1][r / 2], p[2][r / 2]);
            }
            if (A128) {
                uint16_t* d = As + (4 * a_kc + j) * WX_LDK + ((a_mg ^ sw) * 8);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<u32x4v*>(d + pl * BKo * WX_LDK) = u32x4v{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
            } else {
                uint16_t* d = As + (4 * a_kc + j) * WX_LDK + (((a_mg >> 1) ^ sw) * 8) + (a_mg & 1) * 4;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(d + pl * BKo * WX_LDK) = make_uint2(p[pl][0], p[pl][1]);
            }
        }
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            if (!b_on[q]) continue;
            const int swb = (b_nc[q] >> 2) & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = j == 0 ? rb[q][0].x : j == 1 ? rb[q][0].y : j == 2 ? rb[q][0].z : rb[q][0].w;
                const float x1 = j == 0 ? rb[q][1].x : j == 1 ? rb[q][1].y : j == 2 ? rb[q][1].z : rb[q][1].w;
                bsum[q][j] += x0 + x1;
                uint32_t p0, p1, p2;
                split2_bf16x3(x0, x1, p0, p1, p2);
                uint16_t* d = Bs + (4 * b_nc[q] + j) * WX_LDK + ((b_mo[q] ^ swb) * 8) + b_ro[q] * 2;
                *reinterpret_cast<uint32_t*>(d) = p0;
                *reinterpret_cast<uint32_t*>(d + BNo * WX_LDK) = p1;
                *reinterpret_cast<uint32_t*>(d + 2 * BNo * WX_LDK) = p2;
            }
        }
    };
    auto mfma_block = [&]() {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int row = (wm * 2 + a) * 32 + i;
                const int oct = (kb * 2 + h) ^ ((row >> 4) & 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fa[a][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BKo + row) * WX_LDK + oct * 8);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int row = (wn * 2 + b) * 32 + i;
                const int oct = (kb * 2 + h) ^ ((row >> 4) & 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fb[b][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BNo + row) * WX_LDK + oct * 8);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {       // small terms first (order of gemm_x6_kernel)
                    if (kCross21) {
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
        }
    };
    fetch(0);
    for (int t = 0; t < nsteps; ++t) {
        __syncthreads();                       // previous step's fragment reads are done
        swrite();
        __syncthreads();
        fetch(t + 1);
        __builtin_amdgcn_s_setprio(1);         // the MFMA block outranks the co-resident workgroup's staging pass
        __builtin_amdgcn_sched_barrier(0);     // as in gemm_x6_kernel: keep the split arithmetic behind the MFMA block
        mfma_block();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
    }
    // ---- partial slab: dW tile (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    float* ps = part + (long)s * slab;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + (wn * 2 + b) * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < K && col < N) ps[(long)row * N + col] = acc[a][b][r];
            }
        }
    // ---- bias column sums (k tile 0): the 16 (octet, row pair) contributions of a column are combined in fixed order
    if (kt == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(wxs);        // [16][BNo]
#pragma unroll
        for (int q = 0; q < NQB; ++q)
            if (b_on[q])
#pragma unroll
                for (int j = 0; j < 4; ++j) red[(b_mo[q] * 4 + b_ro[q]) * BNo + 4 * b_nc[q] + j] = bsum[q][j];
        __syncthreads();
        for (int n = tid; n < BNo; n += NT) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) sum += red[g * BNo + n];
            if (n0 + n < N) ps[(long)K * N + n0 + n] = sum;
        }
    }
}

struct WgradX8Plan { int cfg = 0, ktiles = 0, ntiles = 0, nslab = 0, steps_per_slab = 0; };
// cfg 1: 256 x 64 tiles (K % 256 == 0, N == 64), 2: 192 x 64 (K % 192 == 0, N == 64), 3: 128 x 128 (N % 128 == 0)
inline WgradX8Plan wgrad_x8_plan(long M, int K, int N, int num_cus, size_t part_floats, bool narrow_too) {
    WgradX8Plan p;
    // N = 64 (conv2 / conv3): 8.1 / 6.9 ms against 5.9 / 4.0 ms for the image-resident fp32-MFMA engine (same box) -- the split
    // engine needs > 0.375 of the bf16 pipe to beat 0.75 of the fp32 pipe, and the 64-wide tiles stay below that
    if (N < 128 && !narrow_too) return p;
    if (M < 1024 || M > 0x7fffffffL || K % 4 != 0 || N % 4 != 0) return p;
    int bk, bn;
    if (N == 64 && K % 256 == 0) { p.cfg = 1; bk = 256; bn = 64; }
    else if (N == 64 && K % 192 == 0) { p.cfg = 2; bk = 192; bn = 64; }
    else if (N % 128 == 0 && K >= 128) { p.cfg = 3; bk = 128; bn = 128; }
    else return p;
    p.ktiles = (K + bk - 1) / bk;
    p.ntiles = N / bn;
    const int T = p.ktiles * p.ntiles;
    const long slab = (long)K * N + N;
    const long steps = (M + 31) / 32;
    long S = std::max<long>(8, (2L * num_cus / T) / 8 * 8);
    S = std::min<long>(S, (long)(part_floats / slab) / 8 * 8);
    S = std::min<long>(S, steps / 4);
    if (S < 1) { p.cfg = 0; return p; }
    p.steps_per_slab = (int)((steps + S - 1) / S);
    p.nslab = (int)((steps + p.steps_per_slab - 1) / p.steps_per_slab);
    return p;
}

template <class AF, int WM, int WN, int NT>
inline hipError_t launch_wgrad_x8_cfg(const AF& af, const float* dz, float* part, long slab, int M, int K, int N, const WgradX8Plan& p,
                                      hipStream_t stream) {
    const size_t lds = (size_t)3 * (WM * 64 + WN * 64) * WX_LDK * sizeof(uint16_t);
    auto kern = wgrad_x8_kernel<AF, WM, WN, NT>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    const int T = p.ktiles * p.ntiles;
    const long blocks = (long)((p.nslab + 7) / 8) * T * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, stream, af, dz, part, slab, M, K, N, p.ktiles, p.ntiles, p.nslab,
                       p.steps_per_slab);
    return hipGetLastError();
}
template <class AF>
inline hipError_t launch_wgrad_x8(const AF& af, const float* dz, float* part, long slab, int M, int K, int N, const WgradX8Plan& p,
                                  hipStream_t stream) {
    switch (p.cfg) {
    case 1: return launch_wgrad_x8_cfg<AF, 4, 1, 256>(af, dz, part, slab, M, K, N, p, stream);
    case 2: return launch_wgrad_x8_cfg<AF, 3, 1, 192>(af, dz, part, slab, M, K, N, p, stream);
    case 3: return launch_wgrad_x8_cfg<AF, 2, 2, 256>(af, dz, part, slab, M, K, N, p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mrl
