// One v_mfma_f32_32x32x16_bf16 onto a random accumulator: error of the result against the exactly rounded sum, in ulps of the result, as a
// function of the size of the 16 products relative to the accumulator (the six product classes of the split arithmetic sit at ~1, 2^-8
// and 2^-16 of a main product; the accumulator is 10 .. 100 main products).  An ideal fused operation has mean 0, rms 0.289.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/mfma_bias_probe.bin scripts/mfma_bias_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ float unif(uint32_t s) { return (hash(s) >> 8) * (1.f / 16777216.f); }            // [0, 1)
__device__ float bf16_round(float x) { return __uint_as_float((__float_as_uint(x) + 0x8000u) & 0xffff0000u); }

// stats[0..3]: sum err, sum err * sign(acc), sum err^2, count   (err in ulps of the exact result's binade)
__global__ void run(float scale, int signmode, double* stats, int f32pipe) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    __shared__ float As[32][16], Bs[32][16];
    double s0 = 0, s1 = 0, s2 = 0;
    for (int rep = 0; rep < 256; ++rep) {
        const uint32_t seed = (blockIdx.x * 256 + rep) * 4096u;
        __syncthreads();
        for (int e = lane; e < 512; e += 64) {
            float a = bf16_round(0.5f + unif(seed + e)), b = bf16_round((0.5f + unif(seed + 1024 + e)) * scale);
            if (signmode == 0 && (hash(seed + 2048 + e) & 1)) b = -b;        // 0: random signs; 1: all positive; 2: all negative
            if (signmode == 2) b = -b;
            As[e >> 4][e & 15] = a; Bs[e >> 4][e & 15] = b;
        }
        __syncthreads();
        uint16_t pa[8], pb[8];
        for (int j = 0; j < 8; ++j) { pa[j] = __float_as_uint(As[i][8 * h + j]) >> 16; pb[j] = __float_as_uint(Bs[i][8 * h + j]) >> 16; }
        bf16x8 fa, fb;
        memcpy(&fa, pa, 16); memcpy(&fb, pb, 16);
        f32x16 acc, acc0;
        for (int r = 0; r < 16; ++r) {
            float v = 1.f + 7.f * unif(seed + 3072 + r * 64 + lane);
            if (hash(seed + 3072 + 1024 + r * 64 + lane) & 1) v = -v;
            acc0[r] = acc[r] = v;
        }
        if (f32pipe) {
            for (int kk = 0; kk < 16; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[i][kk + h], Bs[i][kk + h], acc, 0, 0, 0);
        } else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            double ex = acc0[r];
            for (int k = 0; k < 16; ++k) ex += (double)As[row][k] * (double)Bs[i][k];
            int e2; frexp(ex, &e2);
            const double ulp = ldexp(1.0, e2 - 24);
            const double err = ((double)acc[r] - ex) / ulp;
            s0 += err; s1 += err * (acc0[r] >= 0 ? 1 : -1); s2 += err * err;
        }
    }
    atomicAdd(&stats[0], s0); atomicAdd(&stats[1], s1); atomicAdd(&stats[2], s2); atomicAdd(&stats[3], 256.0 * 16);
}

int main() {
    double* st; hipMalloc(&st, 32);
    const char* sm[3] = {"random signs", "all positive", "all negative"};
    for (int pipe = 0; pipe < 2; ++pipe)
    for (int signmode = 0; signmode < 3; ++signmode) {
        printf("%s, products: %s (accumulator: random sign, |acc| in [1, 8))\n", pipe ? "v_mfma_f32_32x32x2_f32 x 8" : "v_mfma_f32_32x32x16_bf16", sm[signmode]);
        for (int lg = 0; lg >= -28; lg -= 4) {
            hipMemset(st, 0, 32);
            hipLaunchKernelGGL(run, dim3(64), dim3(64), 0, 0, ldexpf(1.f, lg), signmode, st, pipe);
            double h[4]; hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
            printf("  product / acc ~ 2^%-4d mean err %+.4f ulp   mean err*sign(acc) %+.4f   rms %.4f\n", lg - 2, h[0] / h[3], h[1] / h[3], sqrt(h[2] / h[3]));
        }
    }
    return 0;
}
