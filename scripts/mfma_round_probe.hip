// How does v_mfma_f32_32x32x16_bf16 round when it adds its products to the fp32 accumulator?  One non-zero product p per output
// (a[k=0] = 2^-12, b[k=0] = s * 2^-12, everything else 0) is added to acc = 1.0 (ulp 2^-23):
//   p = 0.75 ulp: round-to-nearest gives 1 + ulp, truncation (toward zero) gives 1.0;   p = -0.25 ulp: RN 1.0, RZ 1 - ulp/2 ...
// and the same with v_mfma_f32_32x32x2_f32 (documented as the fmaf chain) for comparison.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/mfma_round_probe.bin scripts/mfma_round_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ uint16_t bf16_bits(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }

__global__ void probe(float acc0, float a0, float b0, float a1, float b1, float* out) {
    const int lane = threadIdx.x;
    // lane (i = lane & 31, kgrp = lane >> 5) holds k = 8 kgrp .. 8 kgrp + 7 of row / column i; put the products at k = 0 and k = 1
    uint16_t av[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) { av[0] = bf16_bits(a0); bv[0] = bf16_bits(b0); av[1] = bf16_bits(a1); bv[1] = bf16_bits(b1); }
    bf16x8 a, b;
    memcpy(&a, av, 16); memcpy(&b, bv, 16);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = acc0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
    f32x16 acc2;
    for (int r = 0; r < 16; ++r) acc2[r] = acc0;
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(lane < 32 ? a0 : a1, lane < 32 ? b0 : b1, acc2, 0, 0, 0);
    if (lane == 0) out[1] = acc2[0];
}

// 16 equal products of `each` ulps (k = 0 .. 15 all populated) added to acc0: is the SUM formed exactly before the one rounding, or is
// every addend first cut to the accumulator's alignment window?
__global__ void probe16(float acc0, float a, float b, float* out, int n = 16) {
    const int lane = threadIdx.x;
    uint16_t av[8], bv[8];
    for (int j = 0; j < 8; ++j) { const bool on = (lane >> 5) * 8 + j < n; av[j] = on ? bf16_bits(a) : 0; bv[j] = on ? bf16_bits(b) : 0; }
    bf16x8 A, B;
    memcpy(&A, av, 16); memcpy(&B, bv, 16);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = acc0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

int main() {
    float* out;
    hipMalloc(&out, 8);
    const float ulp = 1.1920929e-07f;                  // 2^-23
    struct { const char* what; float acc, a0, b0, a1, b1; } cases[] = {
        {"1 + 0.75 ulp            (RN: 1+ulp, RZ: 1)", 1.f, 0.000244140625f, 0.000244140625f * 1.5f, 0.f, 0.f},
        {"1 + 0.25 ulp            (RN: 1, RZ: 1)", 1.f, 0.000244140625f, 0.000244140625f * 0.5f, 0.f, 0.f},
        {"1 - 0.125 ulp           (RN: 1, RZ: 1-ulp/2)", 1.f, 0.000244140625f, -0.000244140625f * 0.25f, 0.f, 0.f},
        {"1 + 0.375 + 0.375 ulp   (two products: exact sum 0.75 ulp)", 1.f, 0.000244140625f, 0.000244140625f * 0.75f, 0.000244140625f, 0.000244140625f * 0.75f},
        {"-1 - 0.75 ulp           (RN: -1-ulp, RZ: -1)", -1.f, 0.000244140625f, -0.000244140625f * 1.5f, 0.f, 0.f},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c.acc, c.a0, c.b0, c.a1, c.b1, out);
        float h[2];
        hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
        printf("%-62s bf16 MFMA: %+.1f ulp   fp32 MFMA: %+.1f ulp\n", c.what, (h[0] - c.acc) / ulp, (h[1] - c.acc) / ulp);
    }
    // a = 2^-12, b = f * 2^-12 * ... : product = f * 2^-24 = f/2 ulp(1.0)
    const float fr[] = {1.f, 0.5f, 0.375f, 0.25f, 0.1875f, 0.125f, 0.0625f, 0.03125f, -1.f, -0.5f, -0.375f, -0.25f, -0.125f};
    for (float f : fr) {
        hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, 1.f, 0.000244140625f, 0.000244140625f * f, out);
        float h;
        hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
        printf("1 + 16 x %.5f ulp (exact sum %.4f ulp; RN of the exact sum: %+.0f ulp)   bf16 MFMA: %+.1f ulp\n", f / 2, 8 * f, rintf(8 * f), (h - 1.f) / ulp);
    }
    for (float f : fr) {
        hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, 1024.f, 0.000244140625f, 0.000244140625f * f * 1024.f, out);
        float h;
        hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
        printf("1024 + 16 x %.5f ulp(1024)  (exact sum %.4f ulp)   bf16 MFMA: %+.1f ulp\n", f / 2, 8 * f, (h - 1024.f) / (ulp * 1024.f));
    }
    // table: n equal addends of f ulp(acc) each, both signs, three accumulators; the result in ulps of the accumulator's binade
    const float accs[] = {1.f, 1.75f, -1.f};
    const int ns[] = {1, 2, 4, 16};
    const float fs[] = {0.25f, 0.1875f, 0.125f, 0.09375f, 0.0625f, 0.046875f, 0.03125f, 0.015625f};
    for (float acc0 : accs)
        for (int sg = 1; sg >= -1; sg -= 2) {
            printf("acc = %+.2f, addends of %s sign: rows n = 1 2 4 16 addends, columns f =", acc0, sg > 0 ? "positive" : "negative");
            for (float f : fs) printf(" %.4f", f);
            printf(" ulp each; entries: result - acc in ulp (exact: n f)\n");
            for (int n : ns) {
                printf("  n=%2d:", n);
                for (float f : fs) {
                    hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, acc0, 0.000244140625f, 0.000244140625f * 2.f * f * sg, out, n);
                    float h;
                    hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
                    printf(" %+7.4f", (h - acc0) / ulp);
                }
                printf("\n");
            }
        }
    return 0;
}
