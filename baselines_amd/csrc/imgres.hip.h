// "Image-resident" fp32-MFMA weight-gradient kernel for gfx950 (MI355X): the third GEMM engine of
// libmrl, for the conv layers' dW / db (tf.gradients of a2c/utils.py:37-56 `conv`).
//
//   dW[(ky,kx,c)][n] = sum over (b, oy, ox) of  X[b, oy*s+ky, ox*s+kx, c] * dz[b, oy, ox, n]
//   db[n]            = sum over (b, oy, ox) of  dz[b, oy, ox, n]
//
// The reduction dimension (batch x pixels) is huge and both operands are streamed once, the output is
// tiny (256x32 ... 576x64).  The tiled kernel spends its time gathering 4-byte im2col fragments from
// global memory (c1.wgrad: 41 % MFMA utilisation, profiles/r01b_pmc_*).  Here instead:
//   * one persistent workgroup per CU owns whole IMAGES: the raw input image (28 KB u8 / 51 KB f32) and
//     its dz map are copied into LDS with fully coalesced 16-byte loads -- 160 KB of LDS holds two
//     such stages, so the next image streams in (global -> VGPR -> LDS) while the current one is consumed;
//   * im2col happens in the LDS ADDRESS: conv-k tile t of 32 rows is 32 consecutive elements of an input
//     row, so the MFMA A operand of lane (i, h) for pixel p is Xs[rowbase(p) + i] -- one conflict-free
//     ds_read (u8 or b32) per MFMA with a compile-time offset, no index arithmetic at run time
//     (geometry is a template parameter; the whole pixel loop is unrolled);
//   * every wave keeps its TMW x TNW accumulator tiles in registers for the kernel's lifetime; one barrier
//     per image; partial sums leave as one slab per workgroup, combined in fixed order by reduce_slabs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hip.h"

namespace mrl {

template <bool U8, int H, int W, int C, int RF, int STRIDE, int NF, int WAVES, int TMW, int TNW>
struct ImgResCfg {
    static constexpr int OH = (H - RF) / STRIDE + 1, OW = (W - RF) / STRIDE + 1;
    static constexpr int NPIX = OH * OW, NSTEP = (NPIX + 1) / 2;
    static constexpr int ROWK = RF * C, K = RF * RF * C;
    static constexpr int MT = K / 32, NTN = NF / 32;
    static constexpr int NT = WAVES * 64;
    static constexpr int XBYTES = (U8 ? 1 : 4) * H * W * C;
    static constexpr int XV = XBYTES / 16;                         // 16-byte vectors per image
    static constexpr int DZV = NPIX * NF / 4;                      // float4 per dz map
    static constexpr int DZ_FLOATS = (NPIX + (NPIX & 1)) * NF;     // + one zero pixel when NPIX is odd
    static constexpr int STAGE_BYTES = XBYTES + DZ_FLOATS * 4;
    static constexpr int NXV = (XV + NT - 1) / NT, NDV = (DZV + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE_BYTES;
    static_assert(K % 32 == 0 && ROWK % 32 == 0 && NF % 32 == 0, "32-wide MFMA tiles");
    static_assert(WAVES * TMW * TNW == MT * NTN, "waves x tiles must cover the output exactly");
    static_assert(NTN % TNW == 0 && XBYTES % 16 == 0 && (NPIX * NF) % 4 == 0 && (NT * 4) % NF == 0, "layout");
    static_assert(LDS_BYTES <= 160 * 1024, "two stages must fit the CU's LDS");
    // element offset of output pixel p's patch origin inside the image
    static constexpr int a_off(int p) { return ((p / OW) * STRIDE * W + (p % OW) * STRIDE) * C; }
};

template <bool U8, int H, int W, int C, int RF, int STRIDE, int NF, int WAVES, int TMW, int TNW, int NACC>
__global__ __launch_bounds__(WAVES * 64) void imgres_wgrad_kernel(const void* __restrict__ x,
                                                                   const int32_t* __restrict__ srow,
                                                                   const float* __restrict__ dz,
                                                                   const float* __restrict__ hcur, int B,
                                                                   float* __restrict__ part) {
    using G = ImgResCfg<U8, H, W, C, RF, STRIDE, NF, WAVES, TMW, TNW>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // this wave's tiles: TNW consecutive n tiles x TMW consecutive m tiles
    constexpr int NGRP = G::NTN / TNW;                 // n groups
    const int mt0 = (wave / NGRP) * TMW, nt0 = (wave % NGRP) * TNW;

    // NACC independent accumulator sets (pixel steps are dealt round-robin, summed at the end): dependent
    // MFMAs on ONE accumulator do not sustain the 64-cycle issue rate, a wave needs >= 2-4 chains
    f32x16 acc[NACC][TMW][TNW];
#pragma unroll
    for (int u = 0; u < NACC; ++u)
#pragma unroll
        for (int a = 0; a < TMW; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][a][b][r] = 0.f;
    float4 bias4 = f4zero();                           // this thread's 4 dz columns, all its loads

    // zero the pad pixel of both stages once
    if (G::NPIX & 1) {
        for (int e = tid; e < 2 * NF; e += G::NT) {
            int st = e / NF, n = e - st * NF;
            reinterpret_cast<float*>(lds + st * G::STAGE_BYTES + G::XBYTES)[G::NPIX * NF + n] = 0.f;
        }
    }

    // ---- streaming copy of the NEXT image into the other stage, one 16-byte vector per thread per
    // "chunk" (chunks 0..NXV-1: image, NXV..NCHUNK-1: dz map).  A chunk is loaded into one of two
    // register slots and written to LDS two chunk-slots later, so only 8 VGPRs are held across the MFMA
    // stream (holding the whole 79 KB stage in registers made the compiler spill and serialise the loads).
    constexpr int NCHUNK = G::NXV + G::NDV;
    // hcur != nullptr: dz is the gradient w.r.t. this layer's OUTPUT; the ReLU mask (hcur > 0) is applied
    // while the chunk is written to LDS (optional mask-in of the incoming gradient; the callers pass hcur == nullptr)
    const uint4* gx = nullptr;
    const float4* gd = nullptr;
    const float4* gh = nullptr;
    auto set_src = [&](int bb) {
        const long img = srow ? (long)srow[bb] : (long)bb;
        gx = reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(x) + img * G::XBYTES);
        gd = reinterpret_cast<const float4*>(dz + (long)bb * G::NPIX * NF);
        gh = reinterpret_cast<const float4*>(hcur + (long)bb * G::NPIX * NF);
    };
    auto load_chunk = [&](int c) -> uint4 {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (c < G::NXV) {
            const int e = tid + c * G::NT;
            if (e < G::XV) v = gx[e];
        } else {
            const int e = tid + (c - G::NXV) * G::NT;
            if (e < G::DZV) {
                const float4 f = gd[e];
                v = make_uint4(__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w));
            }
        }
        return v;
    };
    auto load_mask = [&](int c) -> float4 {          // this layer's activations for a dz chunk (mask source)
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (hcur && c >= G::NXV) {
            const int e = tid + (c - G::NXV) * G::NT;
            if (e < G::DZV) m = gh[e];
        }
        return m;
    };
    auto store_chunk = [&](int c, const uint4& v0, const float4& m, int st) {
        uint4 v = v0;
        if (hcur && c >= G::NXV) {
            v.x = m.x > 0.f ? v.x : 0u; v.y = m.y > 0.f ? v.y : 0u; v.z = m.z > 0.f ? v.z : 0u; v.w = m.w > 0.f ? v.w : 0u;
        }
        if (c < G::NXV) {
            const int e = tid + c * G::NT;
            if (e < G::XV) reinterpret_cast<uint4*>(lds + st * G::STAGE_BYTES)[e] = v;
        } else {
            const int e = tid + (c - G::NXV) * G::NT;
            if (e < G::DZV) reinterpret_cast<uint4*>(lds + st * G::STAGE_BYTES + G::XBYTES)[e] = v;
            bias4.x += __uint_as_float(v.x); bias4.y += __uint_as_float(v.y);     // zeros beyond the map
            bias4.z += __uint_as_float(v.z); bias4.w += __uint_as_float(v.w);
        }
    };

    int b = blockIdx.x;
    if (b < B) {                                       // first image: plain copy
        set_src(b);
        for (int c = 0; c < NCHUNK; ++c) store_chunk(c, load_chunk(c), load_mask(c), 0);
    }
    __syncthreads();
    int stage = 0;
    for (; b < B; b += gridDim.x) {
        const int bn = b + gridDim.x;
        const bool more = bn < B;
        if (more) set_src(bn);
        const uint8_t* xs8 = lds + stage * G::STAGE_BYTES;
        const float* xs32 = reinterpret_cast<const float*>(xs8);
        const float* dzs = reinterpret_cast<const float*>(lds + stage * G::STAGE_BYTES + G::XBYTES) + nt0 * 32 + i + h * NF;
        // per-lane element base of each m tile: conv-k tile t starts at (ky = t*32/ROWK, in-row offset (t*32)%ROWK)
        int abase[TMW];
#pragma unroll
        for (int a = 0; a < TMW; ++a) {
            const int t = mt0 + a;
            abase[a] = ((t * 32) / G::ROWK) * W * C + (t * 32) % G::ROWK + i;
        }
        // Explicit operand pipeline, D steps deep: the ds_reads of step s+D are issued before the MFMAs of
        // step s (and, for u8, the /255 conversion of step s+1 runs under them); sched_barrier pins it.
        constexpr int D = 4;
        uint32_t raw[D][TMW];                 // u8: the byte (converted one step before use);  f32: operand bits
        float fb[D][TNW];
        float cur[TMW];
        uint4 creg[2];
        float4 mreg[2];
        // xo / dzo: element offsets of the current block (image row pair) -- 0 for the fully unrolled form
        auto issue = [&](int s, int slot, int xo, int dzo) {
            const int A0 = G::a_off(2 * s < G::NPIX ? 2 * s : 0);
            const int A1 = (2 * s + 1 < G::NPIX) ? G::a_off(2 * s + 1) : A0;
            const int ao = (h ? A1 : A0) + xo;
#pragma unroll
            for (int a = 0; a < TMW; ++a) {
                if constexpr (U8) raw[slot][a] = xs8[abase[a] + ao];
                else raw[slot][a] = __float_as_uint(xs32[abase[a] + ao]);
            }
#pragma unroll
            for (int c = 0; c < TNW; ++c) fb[slot][c] = dzs[dzo + 2 * s * NF + c * 32];
        };
        auto convert = [&](int slot, float (&dst)[TMW]) {
#pragma unroll
            for (int a = 0; a < TMW; ++a) {
                if constexpr (U8) dst[a] = u8_over_255((float)raw[slot][a]);
                else dst[a] = __uint_as_float(raw[slot][a]);
            }
        };
        // one step: MFMAs of step s (operands in `cur`), conversion of s+1, ds_reads of s+D
        auto step = [&](int s, bool has_next, bool has_ahead, int s_ahead, int xo, int dzo) {
            const int slot = s % D, nslot = (s + 1) % D;
            float fbs[TNW], nxt[TMW];
#pragma unroll
            for (int c = 0; c < TNW; ++c) fbs[c] = fb[slot][c];
            if (has_next) convert(nslot, nxt);
            if (has_ahead) issue(s_ahead, slot, xo, dzo);
#pragma unroll
            for (int a = 0; a < TMW; ++a)
#pragma unroll
                for (int c = 0; c < TNW; ++c)
                    acc[s % NACC][a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[a], fbs[c], acc[s % NACC][a][c], 0, 0, 0);
            if (has_next) {
#pragma unroll
                for (int a = 0; a < TMW; ++a) cur[a] = nxt[a];
            }
            __builtin_amdgcn_sched_barrier(0);
        };

        if constexpr (G::OW % 2 == 0 && G::OH % 2 == 0) {
            // ---- runtime loop over row PAIRS, 2 x OW/2 steps unrolled; pixel offsets are affine in the row
            constexpr int SPR = G::OW / 2, SPI = 2 * SPR;            // steps per row / per iteration
            constexpr int XROW = STRIDE * W * C, DZROW = G::OW * NF;  // element strides of one output row
            static_assert(D <= SPR && SPI % D == 0, "pipeline depth vs row length");
#pragma unroll
            for (int s = 0; s < D; ++s) issue(s, s, 0, 0);
            convert(0, cur);
            for (int q = 0; q < G::OH / 2; ++q) {
                const int xo = q * 2 * XROW, dzo = q * 2 * DZROW;
                const bool last = q + 1 == G::OH / 2;
#pragma unroll
                for (int s = 0; s < SPI; ++s) {
                    if (s == 0 || s == SPR) {                     // chunk slot boundaries: one per row
                        const int cs = s == 0 ? 0 : 1, c = 2 * q + cs;
                        if (more && c >= 2 && c - 2 < NCHUNK) store_chunk(c - 2, creg[cs], mreg[cs], stage ^ 1);
                        if (more && c < NCHUNK) { creg[cs] = load_chunk(c); mreg[cs] = load_mask(c); }
                    }
                    // steps s+1 / s+D may belong to the next iteration: same code, offsets advance by 2 rows
                    const int sa = s + D;
                    if (sa < SPI) step(s, true, true, sa, xo, dzo);
                    else step(s, !last || s + 1 < SPI, !last, sa - SPI, xo + 2 * XROW, dzo + 2 * DZROW);
                }
            }
            if (more) {
#pragma unroll
                for (int c = G::OH; c < NCHUNK + 2; ++c) {           // chunk slots the row loop did not reach
                    if (c - 2 < NCHUNK) store_chunk(c - 2, creg[c & 1], mreg[c & 1], stage ^ 1);
                    if (c < NCHUNK) { creg[c & 1] = load_chunk(c); mreg[c & 1] = load_mask(c); }
                }
            }
        } else {
            // ---- fully unrolled pixel loop (odd map sizes); chunk slots every SP steps
            constexpr int SP = (G::NSTEP - 2) / (NCHUNK + 2) > 0 ? (G::NSTEP - 2) / (NCHUNK + 2) : 1;
#pragma unroll
            for (int s = 0; s < D; ++s)
                if (s < G::NSTEP) issue(s, s, 0, 0);
            convert(0, cur);
#pragma unroll
            for (int s = 0; s < G::NSTEP; ++s) {
                if (s % SP == 0) {
                    const int c = s / SP;
                    if (more && c >= 2 && c - 2 < NCHUNK) store_chunk(c - 2, creg[c & 1], mreg[c & 1], stage ^ 1);
                    if (more && c < NCHUNK) { creg[c & 1] = load_chunk(c); mreg[c & 1] = load_mask(c); }
                }
                step(s, s + 1 < G::NSTEP, s + D < G::NSTEP, s + D, 0, 0);
            }
            if (more) {
                constexpr int CDONE = (G::NSTEP - 1) / SP + 1;       // chunk slots visited inside the loop
#pragma unroll
                for (int c = CDONE; c < NCHUNK + 2; ++c) {
                    if (c - 2 < NCHUNK) store_chunk(c - 2, creg[c & 1], mreg[c & 1], stage ^ 1);
                    if (c < NCHUNK) { creg[c & 1] = load_chunk(c); mreg[c & 1] = load_mask(c); }
                }
            }
        }
        __syncthreads();
        stage ^= 1;
    }

    // ---- partial slab of this workgroup: [K][NF] weights then [NF] bias
    const long slab = (long)G::K * NF + NF;
    float* out = part + (long)blockIdx.x * slab;
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int c = 0; c < TNW; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = (mt0 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                int n = (nt0 + c) * 32 + i;
                float t = acc[0][a][c][r];
#pragma unroll
                for (int u = 1; u < NACC; ++u) t += acc[u][a][c][r];
                out[(long)m * NF + n] = t;
            }
    // bias: thread t owns columns (4t % NF .. +3); combine the NT*4/NF threads of a column in fixed order
    float4* red = reinterpret_cast<float4*>(lds);
    __syncthreads();
    red[tid] = bias4;
    __syncthreads();
    if (tid < NF) {
        constexpr int GROUPS = NF / 4;                 // threads t, t+GROUPS, ... share a column group
        const int g = tid / 4, comp = tid % 4;
        float t = 0.f;
        for (int q = g; q < G::NT; q += GROUPS) {
            const float4 v = red[q];
            t += comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
        }
        out[(long)G::K * NF + tid] = t;
    }
}

template <bool U8, int H, int W, int C, int RF, int STRIDE, int NF, int WAVES, int TMW, int TNW, int NACC>
inline hipError_t launch_imgres_wgrad(const void* x, const int32_t* srow, const float* dz, const float* hcur, int B,
                                      float* part, int nblocks, hipStream_t stream) {
    using G = ImgResCfg<U8, H, W, C, RF, STRIDE, NF, WAVES, TMW, TNW>;
    auto kern = imgres_wgrad_kernel<U8, H, W, C, RF, STRIDE, NF, WAVES, TMW, TNW, NACC>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WAVES * 64), G::LDS_BYTES, stream, x, srow, dz, hcur, B, part);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// First-layer weight gradient on the BF16 matrix pipe ("bf16 x 3", see wres.hip.h): the A operand is raw
// uint8 pixels (exact in bf16), dz is split into three bf16 planes (hi + mid + lo == dz exactly), so every
// product is exact in the fp32 accumulator; 3 v_mfma_f32_32x32x16_bf16 per 16 pixels instead of 8 fp32
// MFMAs per 16 pixels.  The reduction index of this GEMM is the PIXEL, and a bf16 MFMA fragment is 8
// consecutive reduction elements per lane, so both operands need pixel-contiguous runs:
//   * dz is staged fp32 row-major (coalesced), then transposed + split in LDS into dzT[plane][n][pixel]
//     bf16 (a lane reads 8 dz column values, writes three 16-byte runs);
//   * the image stays raw u8: lane (i = byte inside the patch row, g) gathers its 8 pixels with 8 ds_read_u8
//     at compile-time offsets and packs them to bf16.
// dW is accumulated on unscaled pixels and divided by 255 once at the end (models.py:19).
// ------------------------------------------------------------------------------------------
typedef __bf16 ir_bf16x8 __attribute__((ext_vector_type(8)));
struct IrU32x4 { uint32_t x, y, z, w; };

template <int H, int W, int C, int RF, int STRIDE, int NF>
struct ImgResX3Cfg {
    static constexpr int OH = (H - RF) / STRIDE + 1, OW = (W - RF) / STRIDE + 1, NPIX = OH * OW;
    static constexpr int ROWK = RF * C, K = RF * RF * C, MT = K / 32;
    static constexpr int WAVES = MT, NT = WAVES * 64;
    static constexpr int XBYTES = H * W * C, XV = XBYTES / 16;
    static constexpr int DZV = NPIX * NF / 4;
    static constexpr int NPIXP = NPIX + 8;                          // bf16 elements per dzT row
    static constexpr int PLANE = NF * NPIXP;                        // bf16 elements per plane
    static constexpr int OFF_DZF = XBYTES;                          // fp32 staging of dz [NPIX][NF]
    static constexpr int OFF_DZT = OFF_DZF + NPIX * NF * 4;         // 3 bf16 planes
    static constexpr size_t LDS_BYTES = (size_t)OFF_DZT + 3 * (size_t)PLANE * 2;
    static_assert(ROWK == 32 && NF == 32 && NPIX % 16 == 0 && XBYTES % 16 == 0 && MT == RF && NT == 512, "c1-shaped layers only");
    static_assert(LDS_BYTES <= 160 * 1024, "image + dz staging + planes must fit the CU's LDS");
    static constexpr int a_off(int p) { return ((p / OW) * STRIDE * W + (p % OW) * STRIDE) * C; }
};

__device__ __forceinline__ uint32_t ir_bf16_rn_bits(float f) {
    uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <int H, int W, int C, int RF, int STRIDE, int NF>
__global__ __launch_bounds__(512) void imgres_u8x3_wgrad_kernel(
    const void* __restrict__ x, const int32_t* __restrict__ srow, const float* __restrict__ dz, int B,
    float* __restrict__ part) {
    using G = ImgResX3Cfg<H, W, C, RF, STRIDE, NF>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* xs = lds;
    float* dzf = reinterpret_cast<float*>(lds + G::OFF_DZF);
    uint16_t* dzt = reinterpret_cast<uint16_t*>(lds + G::OFF_DZT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 bias4 = f4zero();

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // ---- stage: image bytes and fp32 dz, coalesced 16-byte vectors, loads batched before the LDS writes
        __syncthreads();                                     // previous image fully consumed
        {
            const long img = srow ? (long)srow[b] : (long)b;
            const uint4* gx = reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(x) + img * G::XBYTES);
            const float4* gd = reinterpret_cast<const float4*>(dz + (long)b * G::NPIX * NF);
            constexpr int NXV = (G::XV + G::NT - 1) / G::NT, NDV = (G::DZV + G::NT - 1) / G::NT;
            uint4 vx[NXV];
            float4 vd[NDV];
#pragma unroll
            for (int q = 0; q < NXV; ++q) { const int e = tid + q * G::NT; vx[q] = gx[min(e, G::XV - 1)]; }
#pragma unroll
            for (int q = 0; q < NDV; ++q) { const int e = tid + q * G::NT; vd[q] = gd[min(e, G::DZV - 1)]; }
#pragma unroll
            for (int q = 0; q < NXV; ++q) { const int e = tid + q * G::NT; if (e < G::XV) reinterpret_cast<uint4*>(xs)[e] = vx[q]; }
#pragma unroll
            for (int q = 0; q < NDV; ++q) {
                const int e = tid + q * G::NT;
                if (e < G::DZV) {
                    reinterpret_cast<float4*>(dzf)[e] = vd[q];
                    bias4.x += vd[q].x; bias4.y += vd[q].y; bias4.z += vd[q].z; bias4.w += vd[q].w;
                }
            }
        }
        __syncthreads();
        // ---- transpose + 3-way bf16 split: unit = (column n, run of 8 pixels)
        for (int u = tid; u < NF * (G::NPIX / 8); u += G::NT) {
            const int n = u % NF, grp = u / NF;              // consecutive lanes -> consecutive n: conflict-free reads
            uint32_t pk[3][4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                uint32_t hb[3][2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float v = dzf[(grp * 8 + e2 * 2 + hh) * NF + n];
                    const uint32_t h0 = ir_bf16_rn_bits(v);
                    const float r1 = v - __uint_as_float(h0 << 16);
                    const uint32_t h1 = ir_bf16_rn_bits(r1);
                    const float r2 = r1 - __uint_as_float(h1 << 16);
                    const uint32_t h2 = ir_bf16_rn_bits(r2);
                    hb[0][hh] = h0; hb[1][hh] = h1; hb[2][hh] = h2;
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) pk[pl][e2] = (hb[pl][0] & 0xffffu) | (hb[pl][1] << 16);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<uint4*>(dzt + (long)pl * G::PLANE + (long)n * G::NPIXP + grp * 8) =
                    make_uint4(pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]);
        }
        __syncthreads();
        // ---- MFMA stream: wave = patch row ky; block = 16 pixels (lane half g takes pixels 8g .. 8g+7)
        // launder the lane constant: otherwise all 200 `g ? o1 : o0` offsets are loop-invariant across images
        // and get hoisted out of the image loop (200 live VGPRs -> spills)
        int gq = g;
        asm volatile("" : "+v"(gq));
        const uint8_t* xrow = xs + wave * W * C + i;
        const uint16_t* brow = dzt + (long)i * G::NPIXP + 8 * gq;
        // bytes of block blk+1 are read before the MFMAs of block blk; sched_barrier keeps the compiler from
        // hoisting all 200 byte reads of the unrolled loop (it does, and spills)
        auto rdbytes = [&](int blk, uint32_t (&by)[8]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int o0 = G::a_off(16 * blk + e), o1 = G::a_off(16 * blk + 8 + e);
                by[e] = xrow[gq ? o1 : o0];
            }
        };
        uint32_t by[8], nb[8];
        rdbytes(0, by);
#pragma unroll
        for (int blk = 0; blk < G::NPIX / 16; ++blk) {
            if (blk + 1 < G::NPIX / 16) rdbytes(blk + 1, nb);
            IrU32x4 a;
            {
                const uint32_t f0 = __float_as_uint((float)by[0]), f1 = __float_as_uint((float)by[1]);
                const uint32_t f2 = __float_as_uint((float)by[2]), f3 = __float_as_uint((float)by[3]);
                const uint32_t f4 = __float_as_uint((float)by[4]), f5 = __float_as_uint((float)by[5]);
                const uint32_t f6 = __float_as_uint((float)by[6]), f7 = __float_as_uint((float)by[7]);
                a.x = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
                a.y = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
                a.z = __builtin_amdgcn_perm(f5, f4, 0x07060302u);
                a.w = __builtin_amdgcn_perm(f7, f6, 0x07060302u);
            }
            const ir_bf16x8 av = __builtin_bit_cast(ir_bf16x8, a);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const ir_bf16x8 bv = *reinterpret_cast<const ir_bf16x8*>(brow + (long)pl * G::PLANE + 16 * blk);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
            }
            if (blk + 1 < G::NPIX / 16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) by[e] = nb[e];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- partial slab of this workgroup: [K][NF] weights (scaled by 1/255) then [NF] bias
    const long slab = (long)G::K * NF + NF;
    float* out = part + (long)blockIdx.x * slab;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        out[(long)m * NF + i] = acc[r] / 255.f;
    }
    float4* red = reinterpret_cast<float4*>(lds);
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

template <int H, int W, int C, int RF, int STRIDE, int NF>
inline hipError_t launch_imgres_u8x3_wgrad(const void* x, const int32_t* srow, const float* dz, int B, float* part,
                                           int nblocks, hipStream_t stream) {
    using G = ImgResX3Cfg<H, W, C, RF, STRIDE, NF>;
    auto kern = imgres_u8x3_wgrad_kernel<H, W, C, RF, STRIDE, NF>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(G::NT), G::LDS_BYTES, stream, x, srow, dz, B, part);
    return hipGetLastError();
}

}  // namespace mrl
