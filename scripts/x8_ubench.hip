// Micro-benchmark: what does the inner structure of the tiled split engine (gemmx6.hip.h) sustain on gfx950, piece by piece?
//   hipcc --offload-arch=gfx950 -O3 scripts/x8_ubench.hip -o scripts/x8_ubench.bin && scripts/x8_ubench.bin
// A workgroup = 4 waves, each wave a 64 x 64 tile (2 x 2 accumulators), 8 v_mfma_f32_32x32x16_bf16 per accumulator and
// 16-k block, 2 k blocks per step (BK = 32), operands as 3 bf16 planes in LDS rows of 40 bf16.
//   MODE 0: MFMAs only, operands in registers (accumulator-major: 8 dependent MFMAs in a row)
//   MODE 1: MFMAs only, product-major (the 4 accumulators take turns)
//   MODE 2: + the 12 ds_read_b128 fragment reads per k block (as the engine: reads, then 32 MFMAs)
//   MODE 3: as 2, fragments of the next k block read before the MFMAs of this one (software pipelined)
//   MODE 4: as 2 + the staging pass of the engine per step (split of register data, 24 ds_write_b64 + 3 ds_write_b128,
//           two barriers) -- everything but the global loads
//   MODE 5: as 4 on the pipelined fragment reads of mode 3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
constexpr int BM = 256, BN = 64, LDK = 40;

__device__ __forceinline__ void split2(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    p0 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    p1 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float l0 = r0 - __uint_as_float(v0 & 0xffff0000u), l1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
    for (int e = tid; e < 3 * (BM + BN) * LDK; e += 256) s[e] = (uint16_t)(0x3f80 + (e & 15));
    __syncthreads();
    uint16_t* As = s;
    uint16_t* Bs = s + 3 * BM * LDK;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 ra[8];
    u32x4v rb[3];
    for (int p = 0; p < 8; ++p) ra[p] = make_float4(1.f + tid * 1e-3f, 2.f + p, 3.f, 4.f + lane);
    for (int p = 0; p < 3; ++p) rb[p] = u32x4v{0x3f803f80u, 0x3f803f81u, (uint32_t)tid, 0x3f803f80u};
    auto rd = [&](int kb, bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3]) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[a][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BM + (wave * 2 + a) * 32 + i) * LDK + kb * 16 + 8 * h);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[b][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN + b * 32 + i) * LDK + kb * 16 + 8 * h);
    };
    auto mm = [&](bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3]) {
        if (MODE == 1) {
            constexpr int PA[8] = {2, 1, 2, 1, 0, 1, 0, 0}, PB[8] = {1, 2, 0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[t]], fb[b][PB[t]], acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
                }
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            uint32_t a0x, a1x, a2x, a0y, a1y, a2y;
            split2(ra[p].x, ra[p].y, a0x, a1x, a2x);
            split2(ra[p].z, ra[p].w, a0y, a1y, a2y);
            uint16_t* d = As + (p * 32 + (tid >> 3)) * LDK + (tid & 7) * 4;
            *reinterpret_cast<uint2*>(d) = make_uint2(a0x, a0y);
            *reinterpret_cast<uint2*>(d + BM * LDK) = make_uint2(a1x, a1y);
            *reinterpret_cast<uint2*>(d + 2 * BM * LDK) = make_uint2(a2x, a2y);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4v*>(Bs + (pl * BN + (tid >> 2)) * LDK + (tid & 3) * 8) = rb[pl];
    };
    bf16x8 fa[2][2][3], fb[2][2][3];
    if (MODE <= 1) rd(0, fa[0], fb[0]);
    for (int t = 0; t < steps; ++t) {
        if (MODE >= 4) {
            __syncthreads();
            swrite();
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE <= 1) {
            mm(fa[0], fb[0]);
            mm(fa[0], fb[0]);
        } else if (MODE == 2 || MODE == 4) {
            rd(0, fa[0], fb[0]);
            mm(fa[0], fb[0]);
            rd(1, fa[0], fb[0]);
            mm(fa[0], fb[0]);
        } else {
            rd(0, fa[0], fb[0]);
            rd(1, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            mm(fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            mm(fa[1], fb[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float sum = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE>
void run(const char* name, int wgs_per_cu, float* out) {
    const int steps = 4000, cus = 256;
    const size_t lds = (size_t)3 * (BM + BN) * LDK * 2;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // wgs_per_cu = 1: pad the LDS request so that only one workgroup fits a CU
    const size_t req = wgs_per_cu == 1 ? 100 * 1024 : lds;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(cus * wgs_per_cu), dim3(256), req, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(cus * wgs_per_cu), dim3(256), req, 0, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)cus * wgs_per_cu * 4 * steps * 64 * 32768.0;
    printf("%-58s wg/CU=%d  %7.1f TFLOP/s bf16  (%.3f of 2517)  %.0f cycles/step @2.4GHz\n", name, wgs_per_cu, flops / ms * 1e-9,
           flops / ms * 1e-9 / 2517.0, ms * 1e-3 * 2.4e9 / steps);
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<0>("MFMA only, accumulator-major", w, out);
        run<1>("MFMA only, product-major", w, out);
        run<2>("+ fragment reads (12 b128 then 32 MFMAs)", w, out);
        run<3>("+ fragment reads, both k blocks up front", w, out);
        run<4>("+ staging pass (split, 27 LDS writes, 2 barriers)", w, out);
        run<5>("+ staging pass, fragment reads up front", w, out);
    }
    return 0;
}
