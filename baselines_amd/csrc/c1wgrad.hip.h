// Weight gradient of the first conv layer of NatureCNN (uint8 84x84x4 observations, 32 filters 8x8 stride 4:
// common/models.py:19-21 via a2c/utils.py:37-56; tf.gradients in ppo2/model.py:100-109), image-resident, gfx950:
//     dW[(ky,kx,c)][n] = 1/255 * sum_{b, oy, ox} X[b][4oy+ky][4ox+kx][c] * dz[b][oy][ox][n],   db[n] = sum dz
// on the bf16 matrix pipe with exact products (uint8 pixels are exact in bf16, dz is split exactly into three bf16
// planes; wres.hip.h "bf16 x 3").  The contraction index is the output PIXEL, and v_mfma_f32_32x32x16_bf16 wants 8
// consecutive contraction elements per lane, so BOTH operands are transposed while they are staged:
//   * image: T[y][x & 3][c][x >> 2] (bytes) -- 8 consecutive output pixels of one patch byte (kx, c) are 8 consecutive
//     bytes (two aligned dword reads + v_alignbyte for kx >= 4) instead of 8 single-byte gathers; the 4x4 byte
//     transposes are 8 v_perm per 16 bytes in the staging pass;
//   * dz: threads load 16-byte pieces of 8 consecutive pixels, split, pack PIXEL pairs and write 16-byte runs of
//     dzT[plane][n][pixel] (the transposed-write trick of wgradx8.hip.h), octets XOR-swizzled against bank conflicts;
//   * a wave owns 4 patch rows ky (4 accumulators) and a quarter of the pixels: a dz fragment is read once per 12 MFMAs
//     (the predecessor read it once per 3: 8 waves x the whole dzT per image made LDS the bound);
//   * the next image's pixels and dz are in flight (registers) during the MFMA phase.
// One persistent workgroup per CU, one partial slab per workgroup (summed by reduce_slabs, model.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "c1fwd.hip.h"
#include "gemmx6.hip.h"

namespace mrl {

constexpr int CW_XI = 24;                                // bytes per (y, x&3, c) run: 21 used
constexpr int CW_TROW = 16 * CW_XI;                      // bytes per image row of T
constexpr int CW_TBYTES = C1_H * CW_TROW;                // 32256
constexpr int CW_PP = 432;                               // bf16 per dzT row (400 pixels; 216 dwords = 24 mod 64)
constexpr int CW_PLANE = C1_NF * CW_PP;
constexpr size_t CW_LDS = (size_t)CW_TBYTES + (size_t)3 * CW_PLANE * 2;       // 115200
constexpr int CW_NT = 512;

__global__ __launch_bounds__(CW_NT) void c1wgrad_kernel(const uint8_t* __restrict__ obs, const int32_t* __restrict__ srow,
                                                        const float* __restrict__ dz, int B, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) uint8_t cws[];
    uint8_t* T = cws;
    uint16_t* dzt = reinterpret_cast<uint16_t*>(cws + CW_TBYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int kg = wave & 1, pg = wave >> 1;             // patch rows 4kg .. 4kg+3, pixel quarter

    // ---- staging roles
    const bool img_full = tid < 420, img_small = tid >= 420 && tid < 504;
    const int iy = img_full ? tid / 5 : (img_small ? tid - 420 : 0), iq = img_full ? tid % 5 : 5;
    const int img_off = (iy * C1_W + 16 * iq) * C1_C;    // 16 pixels (64 bytes) of row iy; iq = 5: the last 4 pixels
    const bool dz_on = tid < 400;
    const int d_o = dz_on ? tid >> 3 : 0, d_nc = tid & 7;
    u32x4v vi[4];
    float4 vd[8];
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int b) {
        const long row = srow ? (long)srow[b] : (long)b;
        const uint8_t* gi = obs + row * C1_IMG + img_off;
        vi[0] = *reinterpret_cast<const u32x4v*>(gi);
        if (img_full) {
#pragma unroll
            for (int j = 1; j < 4; ++j) vi[j] = *reinterpret_cast<const u32x4v*>(gi + 16 * j);
        }
        const float* gd = dz + ((long)b * C1_PIX + 8 * d_o) * C1_NF + 4 * d_nc;
#pragma unroll
        for (int r = 0; r < 8; ++r) vd[r] = *reinterpret_cast<const float4*>(gd + r * C1_NF);
    };
    auto stage = [&]() {
        if (img_full) {
            uint8_t* d = T + iy * CW_TROW + 4 * iq;
#pragma unroll
            for (int p = 0; p < 4; ++p) {                 // pixel x = 16 iq + 4 j + p: phase p, run index 4 iq + j
                const uint32_t w0 = vi[0][p], w1 = vi[1][p], w2 = vi[2][p], w3 = vi[3][p];
                const uint32_t a = __builtin_amdgcn_perm(w1, w0, 0x05010400u), bq = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
                const uint32_t c = __builtin_amdgcn_perm(w3, w2, 0x05010400u), dq = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
                *reinterpret_cast<uint32_t*>(d + (p * 4 + 0) * CW_XI) = __builtin_amdgcn_perm(c, a, 0x05040100u);
                *reinterpret_cast<uint32_t*>(d + (p * 4 + 1) * CW_XI) = __builtin_amdgcn_perm(c, a, 0x07060302u);
                *reinterpret_cast<uint32_t*>(d + (p * 4 + 2) * CW_XI) = __builtin_amdgcn_perm(dq, bq, 0x05040100u);
                *reinterpret_cast<uint32_t*>(d + (p * 4 + 3) * CW_XI) = __builtin_amdgcn_perm(dq, bq, 0x07060302u);
            }
        } else if (img_small) {
            uint8_t* d = T + iy * CW_TROW + 20;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[(p * 4 + c) * CW_XI] = (uint8_t)(vi[0][p] >> (8 * c));
        }
        if (dz_on) {
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

    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // ---- MFMA roles: lane (i = kx*4 + c, g): bytes of patch column (kx, c), pixel octet 2 blk + g of the block
    const int kx = i >> 2, cc = i & 3, sh = kx >> 2;
    const uint8_t* tl = T + ((kx & 3) * 4 + cc) * CW_XI + (4 * kg) * CW_TROW;
    const uint16_t* bl = dzt + i * CW_PP;
    const int bsw = i >> 3;
    const int blk0 = pg == 0 ? 0 : 1 + 6 * pg, nblk = pg == 0 ? 7 : 6;       // 25 blocks of 16 pixels: 7 + 6 + 6 + 6

    int b = blockIdx.x;
    if (b < B) fetch(b);
    for (; b < B; b += gridDim.x) {
        __syncthreads();                                   // previous image fully consumed
        stage();
        __syncthreads();
        if (b + (int)gridDim.x < B) fetch(b + gridDim.x);  // next image in flight during the MFMA phase
        __builtin_amdgcn_sched_barrier(0);
        // octet o = 2 blk + g: pixels 8o .. 8o+7; first half at (oyA, oxA), second half (pixel 8o+4) at (oyB, oxB)
        int o = 2 * blk0 + g;
        uint32_t raw[2][4][4];
        bf16x8 bf[2][3];
        auto rd = [&](int oo, uint32_t (&rw)[4][4], bf16x8 (&bb)[3]) {
            const int p0 = 8 * oo, p1 = p0 + 4;
            const int oyA = p0 / C1_OW, oxA = p0 - oyA * C1_OW, oyB = p1 / C1_OW, oxB = p1 - oyB * C1_OW;
            const uint8_t* pa = tl + oyA * (C1_S * CW_TROW) + oxA;
            const uint8_t* pb = tl + oyB * (C1_S * CW_TROW) + oxB;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                rw[a][0] = *reinterpret_cast<const uint32_t*>(pa + a * CW_TROW);
                rw[a][1] = *reinterpret_cast<const uint32_t*>(pa + a * CW_TROW + 4);
                rw[a][2] = *reinterpret_cast<const uint32_t*>(pb + a * CW_TROW);
                rw[a][3] = *reinterpret_cast<const uint32_t*>(pb + a * CW_TROW + 4);
            }
            const uint16_t* bp = bl + ((oo & ~3) | ((oo & 3) ^ bsw)) * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bb[pl] = *reinterpret_cast<const bf16x8*>(bp + pl * CW_PLANE);
        };
        rd(o, raw[0], bf[0]);
        for (int q = 0; q < nblk; q += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (q + u < nblk) {
                    const int cur = u;
                    if (q + u + 1 < nblk) rd(o + 2, raw[cur ^ 1], bf[cur ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const uint32_t lo = __builtin_amdgcn_alignbyte(raw[cur][a][1], raw[cur][a][0], sh);
                        const uint32_t hi = __builtin_amdgcn_alignbyte(raw[cur][a][3], raw[cur][a][2], sh);
                        U32x4 av;
                        u8x4_to_bf16(lo, av.x, av.y);
                        u8x4_to_bf16(hi, av.z, av.w);
                        const bf16x8 af = __builtin_bit_cast(bf16x8, av);
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[cur][2], acc[a], 0, 0, 0);
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[cur][1], acc[a], 0, 0, 0);
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[cur][0], acc[a], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    o += 2;
                }
            }
        }
    }

    // ---- combine the 4 pixel quarters (fixed order) and write the partial slab: [K][NF] weights / 255, then [NF] bias
    __syncthreads();
    float* red = reinterpret_cast<float*>(cws);            // [6 waves][4 acc][16][64 lanes]
    if (pg > 0) {
        float* r0 = red + (long)((pg - 1) * 2 + kg) * 4096;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) r0[(a * 16 + r) * 64 + lane] = acc[a][r];
    }
    __syncthreads();
    const long slab = (long)C1_K * C1_NF + C1_NF;
    float* out = part + (long)blockIdx.x * slab;
    if (pg == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][r];
#pragma unroll
                for (int s = 0; s < 3; ++s) v += red[(long)(s * 2 + kg) * 4096 + (a * 16 + r) * 64 + lane];
                const int m = (4 * kg + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
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

inline hipError_t launch_c1wgrad(const void* obs, const int32_t* srow, const float* dz, int B, float* part, int nblocks,
                                 hipStream_t stream) {
    auto kern = c1wgrad_kernel;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(CW_NT), CW_LDS, stream, static_cast<const uint8_t*>(obs), srow, dz, B, part);
    return hipGetLastError();
}

}  // namespace mrl
