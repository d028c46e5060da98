// Is the six-product bf16 MFMA accumulation biased?  One wave computes a 32 x 32 tile of C = A B^T, K = 3136 (fc1's depth), three ways:
//   x6    the engines' sequence: three exact bf16 planes per operand, six v_mfma_f32_32x32x16_bf16 per 16 k (small classes first)
//   f32   v_mfma_f32_32x32x2_f32
//   x6c   x6 with each class accumulated in its own accumulator and the classes added at the end (fp32 adds)
// against fp64 on the host; prints mean signed error, mean of error * sign(C), rms, all relative to max |C|.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/x6_bias_probe.bin scripts/x6_bias_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 3136;

__device__ __forceinline__ void split3(float x, uint16_t& p0, uint16_t& p1, uint16_t& p2) {
    const uint32_t t0 = __float_as_uint(x) + 0x8000u;
    p0 = t0 >> 16;
    const float r0 = x - __uint_as_float(t0 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0) + 0x8000u;
    p1 = v0 >> 16;
    const float l0 = r0 - __uint_as_float(v0 & 0xffff0000u);
    p2 = __float_as_uint(l0) >> 16;
}

__global__ void tile(const float* A, const float* B, float* out) {   // A [32][K], B [32][K]; out [3][32][32]
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 x6, f32, c0, c1, c2;
    for (int r = 0; r < 16; ++r) x6[r] = f32[r] = c0[r] = c1[r] = c2[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        uint16_t pa[3][8], pb[3][8];
        for (int j = 0; j < 8; ++j) {
            split3(A[i * K + k0 + 8 * h + j], pa[0][j], pa[1][j], pa[2][j]);
            split3(B[i * K + k0 + 8 * h + j], pb[0][j], pb[1][j], pb[2][j]);
        }
        bf16x8 fa[3], fb[3];
        for (int p = 0; p < 3; ++p) { memcpy(&fa[p], pa[p], 16); memcpy(&fb[p], pb[p], 16); }
        x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], x6, 0, 0, 0);
        x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], x6, 0, 0, 0);
        x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], x6, 0, 0, 0);
        x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], x6, 0, 0, 0);
        x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], x6, 0, 0, 0);
        x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], x6, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], c2, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], c0, 0, 0, 0);
        for (int kk = 0; kk < 16; kk += 2)
            f32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + kk + h], B[i * K + k0 + kk + h], f32, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[0 * 1024 + row * 32 + i] = x6[r];
        out[1 * 1024 + row * 32 + i] = f32[r];
        out[2 * 1024 + row * 32 + i] = (c2[r] + c1[r]) + c0[r];
    }
}

int main() {
    const char* names[3] = {"x6 (engine order)", "fp32 MFMA 32x32x2", "x6, one accumulator per class"};
    for (int variant = 0; variant < 3; ++variant) {
        std::mt19937 rng(7 + variant);
        std::normal_distribution<float> nd;
        std::vector<float> A(32 * K), B(32 * K);
        // variant 0: A = relu-like (>= 0, half zero), B mixed sign;  1: both mixed sign;  2: both positive (same-sign accumulation)
        for (auto& v : A) { float x = nd(rng) * 0.1f; v = variant == 1 ? x : (variant == 0 ? (x > 0 ? x : 0.f) : fabsf(x)); }
        for (auto& v : B) { float x = nd(rng) * 0.02f; v = variant == 2 ? fabsf(x) : x; }
        float *dA, *dB, *dO;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 3 * 1024 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(tile, dim3(1), dim3(64), 0, 0, dA, dB, dO);
        std::vector<float> O(3 * 1024);
        hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
        std::vector<double> C(1024);
        double cmax = 0;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * (double)B[n * K + k];
                C[m * 32 + n] = s; cmax = fmax(cmax, fabs(s));
            }
        printf("operands: %s; max |C| = %.3g\n", variant == 0 ? "A >= 0 (ReLU-like), B mixed sign" : variant == 1 ? "both mixed sign" : "both >= 0", cmax);
        for (int e = 0; e < 3; ++e) {
            double mean = 0, msign = 0, sq = 0;
            for (int q = 0; q < 1024; ++q) {
                const double d = ((double)O[e * 1024 + q] - C[q]) / cmax;
                mean += d; msign += d * (C[q] >= 0 ? 1 : -1); sq += d * d;
            }
            printf("  %-32s mean %+.2e   mean(err * sign C) %+.2e   rms %.2e\n", names[e], mean / 1024, msign / 1024, sqrt(sq / 1024));
        }
    }
    return 0;
}
