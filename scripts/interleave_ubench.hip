// Micro-benchmark (round 5): does VALU work HIDE behind v_mfma_f32_32x32x16_bf16 on gfx950 when it sits BETWEEN the MFMAs of
// one wave's instruction stream -- or is SIMD time the SUM of matrix cycles and 4 x VALU instructions, as the phase-separated
// staging pass of scripts/x8_ubench.hip measured (profiles/README.md, round 2)?  And what does the hardware conversion
// v_cvt_pk_bf16_f32 cost next to the 15-instruction integer form of the nearest-rounded 3-way split (wres.hip.h)?
//   hipcc --offload-arch=gfx950 -O3 scripts/interleave_ubench.hip -o scripts/interleave_ubench.bin && scripts/interleave_ubench.bin
// One step of a wave = 48 MFMAs (a 64 x 64 wave tile, 2 k blocks of 16, six plane products) + the split of NP pairs of fp32
// values into three packed bf16 planes; the planes of step t are the A operands of step t+1 and the next inputs depend on the
// last plane (nothing is dead or closed-form).  Program order is pinned with sched_barrier:
//   ORDER 0: no split (matrix pipe alone)    ORDER 1: the whole split, then the 48 MFMAs    ORDER 3: the split alone
//   ORDER 2: one third of a pair's split (a level: 5-7 VALU) after every MFMA (NP = 16), every second (8) ...
//   SPLIT 0: integer rounding, 15 VALU per pair   SPLIT 1: v_cvt_pk_bf16_f32 per level, 11 per pair
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
struct Pair { float x0, x1, r0, r1; uint32_t p0, p1, p2; };
template <int SPLIT> __device__ __forceinline__ void level1(Pair& q) {
    if (SPLIT == 0) {
        const uint32_t t0 = __float_as_uint(q.x0) + 0x8000u, t1 = __float_as_uint(q.x1) + 0x8000u;
        q.p0 = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
        q.r0 = q.x0 - __uint_as_float(t0 & 0xffff0000u); q.r1 = q.x1 - __uint_as_float(t1 & 0xffff0000u);
    } else {
        q.p0 = cvt_pk(q.x0, q.x1);
        q.r0 = q.x0 - __uint_as_float(q.p0 << 16); q.r1 = q.x1 - __uint_as_float(q.p0 & 0xffff0000u);
    }
}
template <int SPLIT> __device__ __forceinline__ void level2(Pair& q) {
    float l0, l1;
    if (SPLIT == 0) {
        const uint32_t v0 = __float_as_uint(q.r0) + 0x8000u, v1 = __float_as_uint(q.r1) + 0x8000u;
        q.p1 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
        l0 = q.r0 - __uint_as_float(v0 & 0xffff0000u); l1 = q.r1 - __uint_as_float(v1 & 0xffff0000u);
    } else {
        q.p1 = cvt_pk(q.r0, q.r1);
        l0 = q.r0 - __uint_as_float(q.p1 << 16); l1 = q.r1 - __uint_as_float(q.p1 & 0xffff0000u);
    }
    q.r0 = l0; q.r1 = l1;
}
__device__ __forceinline__ void level3(Pair& q) {
    q.p2 = __builtin_amdgcn_perm(__float_as_uint(q.r1), __float_as_uint(q.r0), 0x07060302u);
    // next inputs (add + and_or per value, one shift: 5 VALU, counted as overhead of the benchmark)
    q.x0 = __uint_as_float(((__float_as_uint(q.x0) + q.p2) & 0x007fffffu) | 0x3f800000u);
    q.x1 = __uint_as_float(((__float_as_uint(q.x1) + (q.p2 >> 3)) & 0x007fffffu) | 0x3f800000u);
}

template <int ORDER, int SPLIT, int NP>
__global__ __launch_bounds__(256) void k(float* out, int steps, long long* cyc) {
    const int tid = threadIdx.x;
    const long long c0 = clock64();          // s_memtime: shader-clock ticks, whatever clock the power budget leaves
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    Pair q[NP];
    for (int p = 0; p < NP; ++p) { q[p].x0 = 1.f + tid * 1e-3f + p * 0.037f; q[p].x1 = 1.5f + tid * 1e-3f + p * 0.011f; q[p].p0 = q[p].p1 = q[p].p2 = 0x3f803f80u; }
    u32x4v pa[3][4], pb[3][4];
    for (int pl = 0; pl < 3; ++pl)
        for (int j = 0; j < 4; ++j) {
            pa[pl][j] = u32x4v{0x3f803f80u, 0x3f803f81u, (uint32_t)tid, 0x3f803f80u};
            pb[pl][j] = u32x4v{0x3f803f80u, 0x3f813f80u, 0x3f803f80u + j, 0x3f803f80u};
        }
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
    constexpr int EVERY = 48 / (3 * NP) > 0 ? 48 / (3 * NP) : 1;          // a level after every EVERY-th MFMA
    for (int t = 0; t < steps; ++t) {
        if (ORDER == 1 || ORDER == 3) {
#pragma unroll
            for (int p = 0; p < NP; ++p) { level1<SPLIT>(q[p]); level2<SPLIT>(q[p]); level3(q[p]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ORDER != 3) {
#pragma unroll
            for (int m = 0; m < 48; ++m) {
                const int kb = m / 24, pr = (m % 24) / 4, a = (m % 4) / 2, b = m % 2;
                const bf16x8 fa = __builtin_bit_cast(bf16x8, pa[PA[pr]][kb * 2 + a]);
                const bf16x8 fb = __builtin_bit_cast(bf16x8, pb[PB[pr]][kb * 2 + b]);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[a][b], 0, 0, 0);
                if (ORDER == 2) {
                    if (m % EVERY == 0 && m / EVERY < 3 * NP) {
                        const int lv = (m / EVERY) % 3, p = (m / EVERY) / 3;
                        if (lv == 0) level1<SPLIT>(q[p]); else if (lv == 1) level2<SPLIT>(q[p]); else level3(q[p]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // this step's planes are the next step's A operands (16 packed registers per plane)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const Pair& s = q[j % NP];
            pa[0][j / 4][j % 4] = s.p0; pa[1][j / 4][j % 4] = s.p1; pa[2][j / 4][j % 4] = s.p2;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float sum = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
    uint32_t sink = 0;
    for (int p = 0; p < NP; ++p) sink ^= q[p].p0 ^ q[p].p1 ^ q[p].p2;
    out[blockIdx.x * 256 + tid] = sum + __uint_as_float(sink & 0x3fffffffu);
    if (tid == 0) cyc[blockIdx.x] = clock64() - c0;
}

template <int ORDER, int SPLIT, int NP>
void run(const char* name, float* out) {
    for (int w = 1; w <= 2; ++w) {
        const int steps = 4000, cus = 256;
        auto kern = k<ORDER, SPLIT, NP>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const size_t req = w == 1 ? 100 * 1024 : 1024;          // 100 KB of LDS: only one workgroup fits a CU
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        static long long* dcyc = nullptr;
        if (!dcyc) (void)hipMalloc(&dcyc, 1024 * sizeof(long long));
        hipLaunchKernelGGL(kern, dim3(cus * w), dim3(256), req, 0, out, 10, dcyc);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(cus * w), dim3(256), req, 0, out, steps, dcyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long hc[1024];
        (void)hipMemcpy(hc, dcyc, cus * w * sizeof(long long), hipMemcpyDeviceToHost);
        double tot = 0;
        for (int b = 0; b < cus * w; ++b) tot += (double)hc[b];
        const double cyc = tot / (cus * w) / steps / w;            // SHADER cycles per wave-step and SIMD (s_memtime)
        const double ghz = tot / (cus * w) / (ms * 1e-3) * 1e-9;   // the clock the kernel ran at
        const int valu = ORDER == 0 ? 0 : NP * ((SPLIT == 0 ? 15 : 11) + 5);
        printf("%-58s waves/SIMD=%d %6.0f cyc/wave-step (48 MFMAs = 1536; %3d VALU x 4 = %4d; sum %4d)  matrix pipe %.2f  %.2f GHz  %.3f ms\n", name, w,
               cyc, valu, 4 * valu, (ORDER == 3 ? 0 : 1536) + 4 * valu, ORDER == 3 ? 0.0 : 1536.0 / cyc, ghz, ms);
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 1024 * 256 * 4);
    run<0, 0, 16>("matrix pipe alone", out);
    run<3, 0, 16>("VALU alone: integer split, 16 pairs", out);
    run<3, 1, 16>("VALU alone: cvt_pk split, 16 pairs", out);
    run<1, 0, 16>("integer split of 16 pairs, THEN 48 MFMAs", out);
    run<2, 0, 16>("integer split of 16 pairs, a level after every MFMA", out);
    run<1, 1, 16>("cvt_pk split of 16 pairs, THEN 48 MFMAs", out);
    run<2, 1, 16>("cvt_pk split of 16 pairs, a level after every MFMA", out);
    run<1, 0, 8>("integer split of 8 pairs, THEN 48 MFMAs", out);
    run<2, 0, 8>("integer split of 8 pairs, a level after every 2nd MFMA", out);
    run<2, 1, 8>("cvt_pk split of 8 pairs, a level after every 2nd MFMA", out);
    run<2, 0, 4>("integer split of 4 pairs, a level after every 4th MFMA", out);
    return 0;
}
