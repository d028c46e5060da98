// Micro-benchmark: what fp32 MFMA rate do different instruction patterns sustain on gfx950?
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_ubench.hip -o /tmp/mfma_ubench && /tmp/mfma_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// PATTERN 0: NACC independent accumulators, pure MFMA, operands in registers
// PATTERN 1: operands re-read from LDS: one ds_read_b32 pair per MFMA, consumed immediately
// PATTERN 2: as 1 but pipelined 4 MFMAs ahead (explicit ring)
// PATTERN 3: ds_read_b128 pair per 4 MFMAs, prefetched one block ahead
template <int NACC, int PATTERN>
__global__ void k(float* out, int iters) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 8192; e += blockDim.x) lds[e] = 1.0f + 1e-6f * e;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = 1.0f + lane * 1e-3f, y = 0.5f;
    const float* p = lds + lane;
    if (PATTERN == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
        }
    } else if (PATTERN == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    float fa = p[(u * NACC + a) * 64], fb = p[(u * NACC + a) * 64 + 4096];
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    } else if (PATTERN == 2) {
        constexpr int D = 4;
        for (int it = 0; it < iters; ++it) {
            float ra[D], rb[D];
#pragma unroll
            for (int s = 0; s < D; ++s) { ra[s] = p[s * 64]; rb[s] = p[s * 64 + 4096]; }
#pragma unroll
            for (int s = 0; s < 16 * NACC; ++s) {
                float fa = ra[s % D], fb = rb[s % D];
                if (s + D < 16 * NACC) { ra[s % D] = p[((s + D) % 64) * 64]; rb[s % D] = p[((s + D) % 64) * 64 + 4096]; }
                acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[s % NACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        const float4* p4 = reinterpret_cast<const float4*>(lds) + lane;
        for (int it = 0; it < iters; ++it) {
            float4 fa = p4[0], fb = p4[1024];
#pragma unroll
            for (int blk = 0; blk < 4 * NACC; ++blk) {
                float4 na = p4[((blk + 1) % 16) * 64], nb = p4[((blk + 1) % 16) * 64 + 1024];
                const int a = blk % NACC;
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[a], 0, 0, 0);
                fa = na; fb = nb;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int NACC, int PATTERN>
void run(const char* name, int waves_per_cu, float* out) {
    const int iters = 2000, blocks = 256;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto kern = k<NACC, PATTERN>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves_per_cu * 64), 32768, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves_per_cu * 64), 32768, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double mf = (double)blocks * waves_per_cu * iters * 16.0 * NACC * 4096.0;
    printf("%-34s nacc=%d waves/CU=%2d  %7.1f TFLOP/s\n", name, NACC, waves_per_cu, mf / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4 * 4);
    for (int w : {4, 8, 16}) {
        run<1, 0>("pure mfma", w, out);
        run<2, 0>("pure mfma", w, out);
        run<4, 0>("pure mfma", w, out);
        run<1, 1>("ds_read_b32 x2 per mfma, immediate", w, out);
        run<4, 1>("ds_read_b32 x2 per mfma, immediate", w, out);
        run<1, 2>("ds_read_b32 x2 per mfma, ring D=4", w, out);
        run<2, 2>("ds_read_b32 x2 per mfma, ring D=4", w, out);
        run<4, 2>("ds_read_b32 x2 per mfma, ring D=4", w, out);
        run<1, 3>("ds_read_b128 x2 per 4 mfma, pf 1", w, out);
        run<2, 3>("ds_read_b128 x2 per 4 mfma, pf 1", w, out);
        run<4, 3>("ds_read_b128 x2 per 4 mfma, pf 1", w, out);
    }
    return 0;
}
