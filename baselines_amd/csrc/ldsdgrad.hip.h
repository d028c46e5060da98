// LDS-resident fp32-MFMA conv DATA-gradient for gfx950 (MI355X): fourth GEMM engine of libmrl
// (tf.gradients of a2c/utils.py:37-56 `conv` w.r.t. its input, gather form, deterministic).
//
//   dX[b, iy, ix, c] = act'(h_prev) * sum over taps (ky, kx), n of dz[b, (iy-ky)/s, (ix-kx)/s, n] * W[ky, kx, c, n]
//
// As in gemm.hip.h the input pixels are split into s*s stride-parity classes z = (py, px): class rows
// (b, yy, xx) with iy = yy*s + py gather the taps ky = py + s*a only, which makes every class a dense
// GEMM [rows x (taps*NF)] x [(taps*NF) x C].  What is different here (same recipe that took the weight
// gradients from 63 to 106-129 TFLOP/s, profiles/):
//   * a persistent workgroup has ONE role (class z, 32-column slice of C): its slice of the filter bank,
//     33-74 KB, is transposed into LDS once and stays there;
//   * the dz maps of a GROUP of images (5 x 81 or 6 x 49 pixels x 64 ch) are copied into LDS with coalesced
//     16-byte loads (pixel stride padded to 68 floats: conflict-free ds_read_b128 fragments), the next
//     group streams global -> VGPR during the MFMA block;
//   * both MFMA operands are ds_read_b128 fragments with compile-time offsets; taps that fall outside the
//     dz map read a zero pixel, so there is no control flow in the stream; 8 waves x 2 row tiles, explicit
//     fragment prefetch pinned with sched_barrier; two barriers per group of images.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hip.h"

namespace mrl {

template <int H, int W, int C, int RF, int STRIDE, int NF, int G, int WAVES_, int TMW_>
struct LdsDgradCfg {
    static constexpr int OH = (H - RF) / STRIDE + 1, OW = (W - RF) / STRIDE + 1;
    static constexpr int TAPS = (RF + STRIDE - 1) / STRIDE;          // per dimension
    static constexpr int ZC = STRIDE * STRIDE;                       // parity classes
    static constexpr int HY = (H + STRIDE - 1) / STRIDE, WX = (W + STRIDE - 1) / STRIDE, PER = HY * WX;
    static constexpr int NSPL = C / 32;                              // 32-column slices of C
    static constexpr int ROLES = ZC * NSPL;
    static constexpr int KD = TAPS * TAPS * NF, KP = KD + 4;         // class K, padded LDS row
    static constexpr int NFP = NF + 4;                               // padded pixel stride in LDS
    static constexpr int NPIX = OH * OW;
    static constexpr int ROWS = G * PER;                             // class rows per image group
    static constexpr int WAVES = WAVES_, TMW = TMW_, NT = WAVES * 64;
    static constexpr int TILES = WAVES * TMW;
    static constexpr int W_FLOATS = 32 * KP;
    static constexpr int DZ_FLOATS = (G * NPIX + 1) * NFP;           // + the zero pixel
    static constexpr size_t LDS_BYTES = (size_t)(W_FLOATS + DZ_FLOATS) * 4;
    static constexpr int DZV = G * NPIX * NF / 4;                    // float4 per group in global memory
    static constexpr int NDV = (DZV + NT - 1) / NT;
    static_assert(C % 32 == 0 && NF % 8 == 0 && KD % 64 == 0, "tile shapes");
    static_assert(ROWS <= TILES * 32, "an image group must fit the workgroup's 16 row tiles");
    static_assert(27 / WX + 1 < 2 * HY, "epilogue carry arithmetic: a 32-row tile spans at most 3 images");
    static_assert(LDS_BYTES <= 160 * 1024, "filter slice + image group must fit the CU's LDS");
};

template <int H, int W, int C, int RF, int STRIDE, int NF, int G, int WAVES, int TMW>
__global__ __launch_bounds__(WAVES * 64) void lds_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                                const float* __restrict__ hprev, float* __restrict__ out,
                                                                int act, int B, long long* dbg) {
    using K = LdsDgradCfg<H, W, C, RF, STRIDE, NF, G, WAVES, TMW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wt = smem;                                   // [32][KP]   wt[c][tap*NF + n]
    float* dzs = smem + K::W_FLOATS;                    // [G*NPIX + 1][NFP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // All ROLES workgroups that work on the same image groups (same wg) read the same dz bytes at the same time.
    // Consecutive block ids land on consecutive XCDs (8 of them, private L2 each), so the block id is decoded as
    // (xcd, role, slot): the ROLES readers of a group share one XCD's L2 and dz leaves HBM once, not ROLES times.
    const int nwg = gridDim.x / K::ROLES;                                 // host launches a multiple of 8*ROLES
    const int xcd = blockIdx.x % 8, rest = blockIdx.x / 8;
    const int role = rest % K::ROLES, wg = (rest / K::ROLES) * 8 + xcd;
    const int z = role / K::NSPL, cs = role % K::NSPL;
    const int py = z / STRIDE, px = z % STRIDE;
    const int ngroups = (B + G - 1) / G;
    // ---- filter slice -> LDS (once): wt[c][ (a*TAPS + b2)*NF + n ] = W[py + s*a][px + s*b2][cs*32 + c][n]
    for (int e = tid; e < 32 * K::KD; e += K::NT) {
        const int n = e % NF, t = (e / NF) % (K::TAPS * K::TAPS), c = e / (NF * K::TAPS * K::TAPS);
        const int a = t / K::TAPS, b2 = t % K::TAPS;
        const int ky = py + STRIDE * a, kx = px + STRIDE * b2;
        float v = 0.f;
        if (ky < RF && kx < RF) v = w[((long)(ky * RF + kx) * C + cs * 32 + c) * NF + n];
        wt[c * K::KP + t * NF + n] = v;
    }
    for (int e = tid; e < K::NFP; e += K::NT) dzs[(long)G * K::NPIX * K::NFP + e] = 0.f;    // zero pixel

    // The copy is a plain synchronous phase between two barriers (global -> VGPR -> LDS, 4 vectors in flight per
    // thread).  Holding the next group in registers across the MFMA stream instead made the compiler spill and
    // serialise (measured slower); the epilogue's stores of the previous group are still in flight during the copy.
    auto stage_group = [&](int grp) {
        const long base4 = (long)grp * K::DZV;
        const float4* src = reinterpret_cast<const float4*>(dz) + base4;
        const long left4 = (long)B * K::NPIX * NF / 4 - base4;
        const int nvalid = (int)(left4 < (long)K::DZV ? left4 : (long)K::DZV);
        constexpr int U = 4;
        for (int e0 = tid; e0 < K::DZV; e0 += U * K::NT) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = min(e0 + u * K::NT, nvalid - 1);
                v[u] = src[e];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * K::NT;
                if (e < K::DZV) {
                    float4 t = e < nvalid ? v[u] : f4zero();
                    const int pix = (e * 4) / NF, n = (e * 4) % NF;     // NF/4 float4 per pixel
                    *reinterpret_cast<float4*>(dzs + (long)pix * K::NFP + n) = t;
                }
            }
        }
    };

    // dbg != nullptr (timing experiments): wave `dbgw` of workgroup 0 stamps the phase boundaries of its first 6 groups
    int it = 0;
    auto stamp = [&](int k) {
        if (dbg && wg == 0 && role == 0 && lane == 0 && (wave == 0 || wave == WAVES - 1) && it < 6)
            dbg[((wave ? 1 : 0) * 6 + it) * 8 + k] = (long long)__builtin_readcyclecounter();
    };
    for (int grp = wg; grp < ngroups; grp += nwg, ++it) {
        stamp(0);
        __syncthreads();                                 // everyone is done reading the previous group
        stamp(1);
        stage_group(grp);
        stamp(2);
        __syncthreads();
        stamp(3);
        const int b0 = grp * G;

        // ---- this wave's two row tiles; lane row -> (image g, yy, xx) of class z
        int pixbase[K::TMW], yy[K::TMW], xx[K::TMW];
#pragma unroll
        for (int t = 0; t < K::TMW; ++t) {
            const int m = (wave * K::TMW + t) * 32 + i;
            const int g = m / K::PER, r = m % K::PER;
            const bool ok = m < K::ROWS && b0 + g < B;
            yy[t] = ok ? r / K::WX : -0x10000;           // never in range -> zero pixel
            xx[t] = r % K::WX;
            pixbase[t] = g * K::NPIX;
        }
        f32x16 acc[K::TMW];
#pragma unroll
        for (int t = 0; t < K::TMW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const float* wrow = wt + i * K::KP + h * 4;
        constexpr int NBLK = NF / 8;                     // k blocks (8 k) per tap
        constexpr int ZPIX = G * K::NPIX;                // index of the zero pixel
#pragma unroll
        for (int tap = 0; tap < K::TAPS * K::TAPS; ++tap) {
            const int a = tap / K::TAPS, b2 = tap % K::TAPS;
            const float* ap[K::TMW];
#pragma unroll
            for (int t = 0; t < K::TMW; ++t) {
                const int oy = yy[t] - a, ox = xx[t] - b2;
                const bool ok = (unsigned)oy < (unsigned)K::OH && (unsigned)ox < (unsigned)K::OW;
                const int pix = ok ? pixbase[t] + oy * K::OW + ox : ZPIX;
                ap[t] = dzs + (long)pix * K::NFP + h * 4;
            }
            // fragments of block 0, then prefetch block j+1 ahead of block j's MFMAs
            float4 fa[K::TMW], fb;
#pragma unroll
            for (int t = 0; t < K::TMW; ++t) fa[t] = *reinterpret_cast<const float4*>(ap[t]);
            fb = *reinterpret_cast<const float4*>(wrow + tap * NF);
#pragma unroll
            for (int j = 0; j < NBLK; ++j) {
                float4 na[K::TMW], nb;
                if (j + 1 < NBLK) {
#pragma unroll
                    for (int t = 0; t < K::TMW; ++t) na[t] = *reinterpret_cast<const float4*>(ap[t] + (j + 1) * 8);
                    nb = *reinterpret_cast<const float4*>(wrow + tap * NF + (j + 1) * 8);
                }
#pragma unroll
                for (int t = 0; t < K::TMW; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t].x, fb.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t].y, fb.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t].z, fb.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t].w, fb.w, acc[t], 0, 0, 0);
                }
                if (j + 1 < NBLK) {
#pragma unroll
                    for (int t = 0; t < K::TMW; ++t) fa[t] = na[t];
                    fb = nb;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(4);
        // ---- epilogue: C/D layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*h
        // all act'-mask loads of both tiles are issued before the first store (a load -> wait -> store
        // chain per element costs 32 serial memory round trips per group)
        float hv[K::TMW][16];
        // Destination offsets.  The row -> (image, iy, ix) map is invariant across groups: left alone the compiler
        // hoists all 32 offsets out of the group loop (64+ VGPRs live across the MFMA stream -> spills that
        // serialise the prefetch), so the lane constants are laundered and the math stays here -- but it has to
        // be cheap, 4 waves of a SIMD run it back to back while the matrix pipe idles: 32-bit element offsets
        // relative to the group's first image (a group is < 2^31 elements), one div/mod chain per tile and
        // carry arithmetic for its 16 rows (row r of the C/D layout is base + (r&3) + 8*(r>>2)).
        int hq = h, wq = wave;
        asm volatile("" : "+v"(hq), "+v"(wq));
        const long gbase = (long)b0 * (H * W * C) + cs * 32 + i;
        const int gleft = B - b0;                                      // images of this group that exist
        int goff[K::TMW][16];
#pragma unroll
        for (int t = 0; t < K::TMW; ++t) {
            const int m0 = (wq * K::TMW + t) * 32 + 4 * hq;
            const int g0 = m0 / K::PER, r0 = m0 % K::PER;
            const int yy0 = r0 / K::WX, xx0 = r0 % K::WX;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = (r & 3) + 8 * (r >> 2);                  // <= 27
                int xx1 = xx0 + d, yy1 = yy0, g1 = g0;
                // d / WX and d % WX are compile-time constants; at most one extra carry from the remainders
                yy1 += d / K::WX;
                xx1 = xx0 + d % K::WX;
                if (xx1 >= K::WX) { xx1 -= K::WX; ++yy1; }
                if (yy1 >= 2 * K::HY) { yy1 -= 2 * K::HY; g1 += 2; }
                if (yy1 >= K::HY) { yy1 -= K::HY; ++g1; }
                const int iy = yy1 * STRIDE + py, ix = xx1 * STRIDE + px;
                const bool ok = g1 < G && g1 < gleft && iy < H && ix < W;
                goff[t][r] = ok ? ((g1 * H + iy) * W + ix) * C : -1;
            }
        }
        auto off = [&](int t, int r) -> long { return goff[t][r] < 0 ? -1 : gbase + goff[t][r]; };
#pragma unroll
        for (int t = 0; t < K::TMW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long o = off(t, r);
                hv[t][r] = hprev[o < 0 ? 0 : o];
            }
        stamp(5);
#pragma unroll
        for (int t = 0; t < K::TMW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long o = off(t, r);
                if (o >= 0) out[o] = acc[t][r] * act_bwd_from_out(hv[t][r], act);
            }
        stamp(6);
    }
}

// ------------------------------------------------------------------------------------------
// Asynchronous variant (the default): the three memory phases of a group no longer sit between the MFMA streams.
//   * staging is LDS-DMA (`global_load_lds`, 1 KB per wave instruction, no VGPR round trip, no ds_write): issued right
//     after the barrier that ends a group's MFMA phase, it runs while the waves compute the NEXT group's destination
//     offsets.  The DMA image is lane-linear, so the pixel rows cannot be padded; bank conflicts are avoided by an XOR
//     swizzle applied through the SOURCE address (chunk c of pixel p is stored at position c ^ (p & 15)) and undone
//     in the fragment reads;
//   * the act' mask values (h_prev) are loaded BEFORE the MFMA stream into registers (8 waves x 2 row tiles leave
//     room: 256 VGPRs per lane) and the destination offsets are computed once per group and kept -- the epilogue is
//     32 fire-and-forget stores per wave.
// ------------------------------------------------------------------------------------------
template <int H, int W, int C, int RF, int STRIDE, int NF, int G, int WAVES_, int TMW_>
struct LdsDgradAsyncCfg : LdsDgradCfg<H, W, C, RF, STRIDE, NF, G, WAVES_, TMW_> {
    using Base = LdsDgradCfg<H, W, C, RF, STRIDE, NF, G, WAVES_, TMW_>;
    static constexpr int QUADS = (G * Base::NPIX + 3) / 4;           // 4 pixels (1 KB) per DMA instruction
    static constexpr int ZPIX = QUADS * 4;                           // index of the zero pixel
    static constexpr int DZ_FLOATS = (ZPIX + 1) * NF;
    static constexpr size_t LDS_BYTES = (size_t)(Base::W_FLOATS + DZ_FLOATS) * 4;
    static_assert(NF == 64, "a pixel is 16 chunks of 16 bytes (the swizzle domain)");
    static_assert(LDS_BYTES <= 160 * 1024, "filter slice + image group must fit the CU's LDS");
};

template <int H, int W, int C, int RF, int STRIDE, int NF, int G, int WAVES, int TMW>
__global__ __launch_bounds__(WAVES * 64) void lds_dgrad_async_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                                      const float* __restrict__ hprev, float* __restrict__ out,
                                                                      int act, int B, long long* dbg) {
    using K = LdsDgradAsyncCfg<H, W, C, RF, STRIDE, NF, G, WAVES, TMW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wt = smem;                                   // [32][KP]   wt[c][tap*NF + n]
    float* dzs = smem + K::W_FLOATS;                    // [ZPIX + 1][NF], chunk-swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int nwg = gridDim.x / K::ROLES;                                 // host launches a multiple of 8*ROLES
    const int xcd = blockIdx.x % 8, rest = blockIdx.x / 8;
    const int role = rest % K::ROLES, wg = (rest / K::ROLES) * 8 + xcd;   // the ROLES readers of a group share one XCD
    const int z = role / K::NSPL, cs = role % K::NSPL;
    const int py = z / STRIDE, px = z % STRIDE;
    const int ngroups = (B + G - 1) / G;
    for (int e = tid; e < 32 * K::KD; e += K::NT) {
        const int n = e % NF, t = (e / NF) % (K::TAPS * K::TAPS), c = e / (NF * K::TAPS * K::TAPS);
        const int a = t / K::TAPS, b2 = t % K::TAPS;
        const int ky = py + STRIDE * a, kx = px + STRIDE * b2;
        float v = 0.f;
        if (ky < RF && kx < RF) v = w[((long)(ky * RF + kx) * C + cs * 32 + c) * NF + n];
        wt[c * K::KP + t * NF + n] = v;
    }
    for (int e = tid; e < NF; e += K::NT) dzs[(long)K::ZPIX * NF + e] = 0.f;        // zero pixel

    const long pixmax = (long)B * K::NPIX - 1;
    auto issue_stage = [&](int grp) {
        const long pix0 = (long)grp * G * K::NPIX;
        for (int q = wave; q < K::QUADS; q += WAVES) {
            const int pl = 4 * q + (lane >> 4);
            const long gp = min(pix0 + pl, pixmax);              // past the end: duplicates, never referenced
            const float* src = dz + gp * NF + (((lane & 15) ^ (pl & 15)) << 2);
            __builtin_amdgcn_global_load_lds(src, dzs + (long)4 * q * NF, 16, 0, 0);
        }
    };
    int it = 0;
    auto stamp = [&](int k) {
        if (dbg && wg == 0 && role == 0 && lane == 0 && (wave == 0 || wave == WAVES - 1) && it < 6)
            dbg[((wave ? 1 : 0) * 6 + it) * 8 + k] = (long long)__builtin_readcyclecounter();
    };
    if (wg < ngroups) issue_stage(wg);
    for (int grp = wg; grp < ngroups; grp += nwg, ++it) {
        stamp(0);
        const int b0 = grp * G;
        // ---- destination offsets of this wave's tiles (32-bit, relative to the group's first image), kept for the epilogue
        const long gbase = (long)b0 * (H * W * C) + cs * 32 + i;
        const int gleft = B - b0;
        int goff[TMW][16];
#pragma unroll
        for (int t = 0; t < TMW; ++t) {
            const int m0 = (wave * TMW + t) * 32 + 4 * h;
            const int g0 = m0 / K::PER, r0 = m0 % K::PER;
            const int yy0 = r0 / K::WX, xx0 = r0 % K::WX;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = (r & 3) + 8 * (r >> 2);
                int yy1 = yy0 + d / K::WX, g1 = g0;
                int xx1 = xx0 + d % K::WX;
                if (xx1 >= K::WX) { xx1 -= K::WX; ++yy1; }
                if (yy1 >= 2 * K::HY) { yy1 -= 2 * K::HY; g1 += 2; }
                if (yy1 >= K::HY) { yy1 -= K::HY; ++g1; }
                const int iy = yy1 * STRIDE + py, ix = xx1 * STRIDE + px;
                const bool ok = g1 < G && g1 < gleft && iy < H && ix < W;
                goff[t][r] = ok ? ((g1 * H + iy) * W + ix) * C : -1;
            }
        }
        stamp(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this group's DMA has landed
        __syncthreads();
        stamp(2);
        // ---- act' mask values: in flight during the MFMA stream
        float hv[TMW][16];
#pragma unroll
        for (int t = 0; t < TMW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[t][r] = hprev[gbase + (goff[t][r] < 0 ? 0 : goff[t][r])];
        __builtin_amdgcn_sched_barrier(0);

        int pixbase[TMW], yy[TMW], xx[TMW];
#pragma unroll
        for (int t = 0; t < TMW; ++t) {
            const int m = (wave * TMW + t) * 32 + i;
            const int g = m / K::PER, r = m % K::PER;
            const bool ok = m < K::ROWS && b0 + g < B;
            yy[t] = ok ? r / K::WX : -0x10000;
            xx[t] = r % K::WX;
            pixbase[t] = g * K::NPIX;
        }
        f32x16 acc[TMW];
#pragma unroll
        for (int t = 0; t < TMW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const float* wrow = wt + i * K::KP + h * 4;
        constexpr int NBLK = NF / 8;
#pragma unroll
        for (int tap = 0; tap < K::TAPS * K::TAPS; ++tap) {
            const int a = tap / K::TAPS, b2 = tap % K::TAPS;
            const float* ap[TMW];
            int sw[TMW];
#pragma unroll
            for (int t = 0; t < TMW; ++t) {
                const int oy = yy[t] - a, ox = xx[t] - b2;
                const bool ok = (unsigned)oy < (unsigned)K::OH && (unsigned)ox < (unsigned)K::OW;
                const int pix = ok ? pixbase[t] + oy * K::OW + ox : K::ZPIX;
                ap[t] = dzs + (long)pix * NF;
                sw[t] = pix & 15;
            }
            // chunk (2j + h) of pixel p sits at position (2j + h) ^ (p & 15)
            float4 fa[TMW], fb;
#pragma unroll
            for (int t = 0; t < TMW; ++t) fa[t] = *reinterpret_cast<const float4*>(ap[t] + ((h ^ sw[t]) << 2));
            fb = *reinterpret_cast<const float4*>(wrow + tap * NF);
#pragma unroll
            for (int j = 0; j < NBLK; ++j) {
                float4 na[TMW], nb;
                if (j + 1 < NBLK) {
#pragma unroll
                    for (int t = 0; t < TMW; ++t)
                        na[t] = *reinterpret_cast<const float4*>(ap[t] + ((((j + 1) * 2 + h) ^ sw[t]) << 2));
                    nb = *reinterpret_cast<const float4*>(wrow + tap * NF + (j + 1) * 8);
                }
#pragma unroll
                for (int t = 0; t < TMW; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t].x, fb.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t].y, fb.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t].z, fb.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t].w, fb.w, acc[t], 0, 0, 0);
                }
                if (j + 1 < NBLK) {
#pragma unroll
                    for (int t = 0; t < TMW; ++t) fa[t] = na[t];
                    fb = nb;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(3);
        // ---- epilogue: stores only (mask values arrived long ago)
#pragma unroll
        for (int t = 0; t < TMW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (goff[t][r] >= 0) out[gbase + goff[t][r]] = acc[t][r] * act_bwd_from_out(hv[t][r], act);
        stamp(4);
        __syncthreads();                                 // everyone is done reading this group's dz
        stamp(5);
        if (grp + nwg < ngroups) issue_stage(grp + nwg);
        stamp(6);
    }
}

template <int H, int W, int C, int RF, int STRIDE, int NF, int G, int WAVES, int TMW>
inline hipError_t launch_lds_dgrad_async(const float* dz, const float* w, const float* hprev, float* out, int act, int B,
                                         int num_cus, long long* dbg, hipStream_t stream) {
    using K = LdsDgradAsyncCfg<H, W, C, RF, STRIDE, NF, G, WAVES, TMW>;
    auto kern = lds_dgrad_async_kernel<H, W, C, RF, STRIDE, NF, G, WAVES, TMW>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    const int ngroups = (B + G - 1) / G;
    int per_role = std::max(8, std::min(num_cus / K::ROLES, (ngroups + 7) / 8 * 8) / 8 * 8);   // multiple of 8 (XCD decode)
    hipLaunchKernelGGL(kern, dim3(per_role * K::ROLES), dim3(WAVES * 64), K::LDS_BYTES, stream, dz, w, hprev, out, act, B, dbg);
    return hipGetLastError();
}

template <int H, int W, int C, int RF, int STRIDE, int NF, int G, int WAVES, int TMW>
inline hipError_t launch_lds_dgrad(const float* dz, const float* w, const float* hprev, float* out, int act, int B,
                                   int num_cus, long long* dbg, hipStream_t stream) {
    using K = LdsDgradCfg<H, W, C, RF, STRIDE, NF, G, WAVES, TMW>;
    auto kern = lds_dgrad_kernel<H, W, C, RF, STRIDE, NF, G, WAVES, TMW>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    const int ngroups = (B + G - 1) / G;
    int per_role = std::max(8, std::min(num_cus / K::ROLES, (ngroups + 7) / 8 * 8) / 8 * 8);   // multiple of 8 (XCD decode)
    hipLaunchKernelGGL(kern, dim3(per_role * K::ROLES), dim3(WAVES * 64), K::LDS_BYTES, stream, dz, w, hprev, out, act, B, dbg);
    return hipGetLastError();
}

}  // namespace mrl
