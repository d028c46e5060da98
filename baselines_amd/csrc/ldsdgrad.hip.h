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
    static_assert(LDS_BYTES <= 160 * 1024, "filter slice + image group must fit the CU's LDS");
};

template <int H, int W, int C, int RF, int STRIDE, int NF, int G, int WAVES, int TMW, int DBG>
__global__ __launch_bounds__(WAVES * 64) void lds_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ hcur,
                                                                const float* __restrict__ w, const float* __restrict__ hprev,
                                                                float* __restrict__ out, int act, int B) {
    using K = LdsDgradCfg<H, W, C, RF, STRIDE, NF, G, WAVES, TMW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wt = smem;                                   // [32][KP]   wt[c][tap*NF + n]
    float* dzs = smem + K::W_FLOATS;                    // [G*NPIX + 1][NFP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int role = blockIdx.x % K::ROLES, z = role / K::NSPL, cs = role % K::NSPL;
    const int py = z / STRIDE, px = z % STRIDE;
    const int wg = blockIdx.x / K::ROLES, nwg = gridDim.x / K::ROLES;     // host launches a multiple of ROLES
    const int ngroups = (B + G - 1) / G;
    // DBG (timing experiments only): 1 = skip the epilogue, 2 = skip the per-group staging, 3 = both
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

    // Deferred activation mask: when hcur != nullptr the incoming dz is the gradient w.r.t. this layer's
    // OUTPUT (the producer skipped its act' epilogue, whose loads cannot overlap anything there); the ReLU
    // mask (hcur > 0) is applied here, while copying into LDS.
    //
    // The copy is a plain synchronous phase between two barriers (global -> VGPR -> LDS, 4 vectors in
    // flight per thread): ~5 % of a group's time.  Holding the next group in registers across the MFMA
    // stream instead made the compiler spill and serialise (measured slower); the epilogue's stores of the
    // previous group are still in flight during the copy, so the two memory phases overlap each other.
    auto stage_group = [&](int grp) {
        const long base4 = (long)grp * K::DZV;
        const float4* src = reinterpret_cast<const float4*>(dz) + base4;
        const float4* hsrc = reinterpret_cast<const float4*>(hcur) + base4;
        const long left4 = (long)B * K::NPIX * NF / 4 - base4;
        const int nvalid = (int)(left4 < (long)K::DZV ? left4 : (long)K::DZV);
        constexpr int U = 4;
        for (int e0 = tid; e0 < K::DZV; e0 += U * K::NT) {
            float4 v[U], hm[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = min(e0 + u * K::NT, nvalid - 1);
                v[u] = src[e];
                if (hcur) hm[u] = hsrc[e];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * K::NT;
                if (e < K::DZV) {
                    float4 t = e < nvalid ? v[u] : f4zero();
                    if (hcur) {
                        t.x = hm[u].x > 0.f ? t.x : 0.f; t.y = hm[u].y > 0.f ? t.y : 0.f;
                        t.z = hm[u].z > 0.f ? t.z : 0.f; t.w = hm[u].w > 0.f ? t.w : 0.f;
                    }
                    const int pix = (e * 4) / NF, n = (e * 4) % NF;     // NF/4 float4 per pixel
                    *reinterpret_cast<float4*>(dzs + (long)pix * K::NFP + n) = t;
                }
            }
        }
    };

    for (int grp = wg; grp < ngroups; grp += nwg) {
        __syncthreads();                                 // everyone is done reading the previous group
        if (!(DBG & 2) || grp == wg) stage_group(grp);
        __syncthreads();
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
        // ---- epilogue: C/D layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*h
        // all act'-mask loads of both tiles are issued before the first store (a load -> wait -> store
        // chain per element costs 32 serial memory round trips per group)
        if ((DBG & 1) && B > 0) {          // keep the accumulators alive without touching memory
            float t = 0.f;
#pragma unroll
            for (int tt = 0; tt < K::TMW; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[tt][r];
            if (t == 123.456f) out[0] = t;
            continue;
        }
        float hv[K::TMW][16];
        // the row -> (image, iy, ix) part of these offsets is invariant across groups; left alone the compiler
        // hoists all 32 of them out of the group loop (64+ VGPRs live across the MFMA stream -> spills that
        // serialise the prefetch).  Laundering the lane constants keeps the arithmetic in the epilogue.
        int hq = h, wq = wave;
        asm volatile("" : "+v"(hq), "+v"(wq));
        auto off = [&](int t, int r) -> long {
            const int m = (wq * K::TMW + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hq;
            const int g = m / K::PER, rr = m % K::PER;
            const int iy = (rr / K::WX) * STRIDE + py, ix = (rr % K::WX) * STRIDE + px;
            const bool ok = m < K::ROWS && b0 + g < B && iy < H && ix < W;
            return ok ? ((long)((b0 + g) * H + iy) * W + ix) * C + cs * 32 + i : -1;
        };
        if (hprev) {
#pragma unroll
            for (int t = 0; t < K::TMW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long o = off(t, r);
                    hv[t][r] = hprev[o < 0 ? 0 : o];
                }
#pragma unroll
            for (int t = 0; t < K::TMW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long o = off(t, r);
                    if (o >= 0) out[o] = acc[t][r] * act_bwd_from_out(hv[t][r], act);
                }
        } else {                                   // mask deferred to the consumers: fire-and-forget stores
#pragma unroll
            for (int t = 0; t < K::TMW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long o = off(t, r);
                    if (o >= 0) out[o] = acc[t][r];
                }
        }
    }
}

template <int H, int W, int C, int RF, int STRIDE, int NF, int G, int WAVES, int TMW, int DBG>
inline hipError_t launch_lds_dgrad(const float* dz, const float* hcur, const float* w, const float* hprev, float* out,
                                   int act, int B, int num_cus, hipStream_t stream) {
    using K = LdsDgradCfg<H, W, C, RF, STRIDE, NF, G, WAVES, TMW>;
    auto kern = lds_dgrad_kernel<H, W, C, RF, STRIDE, NF, G, WAVES, TMW, DBG>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised = true;
    }
    const int ngroups = (B + G - 1) / G;
    int per_role = std::max(1, std::min(num_cus / K::ROLES, ngroups));
    hipLaunchKernelGGL(kern, dim3(per_role * K::ROLES), dim3(WAVES * 64), K::LDS_BYTES, stream, dz, hcur, w, hprev, out, act, B);
    return hipGetLastError();
}

}  // namespace mrl
