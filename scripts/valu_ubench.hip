// Micro-benchmark: issue cost of the VALU instructions the staging passes are made of (gfx950), 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_ubench.hip -o scripts/valu_ubench.bin && scripts/valu_ubench.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = seed * (threadIdx.x + 1) + j * 0x01010101u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t x = a[j], y = a[(j + 1) & 7];
                if (OP == 0) x = __builtin_amdgcn_perm(x, y, 0x07060302u);
                else if (OP == 1) x = __float_as_uint((float)((x >> 8) & 0xff)) ^ y;      // v_cvt_f32_ubyte1 + xor
                else if (OP == 2) x = __float_as_uint(__uint_as_float(x | 0x3f800000u) - __uint_as_float(y & 0xffff0000u));   // or + and + sub
                else if (OP == 3) x = __builtin_amdgcn_alignbyte(x, y, 2);
                else if (OP == 4) x = x ^ (y + 0x9e3779b9u);                              // xor + add: plain integer ops
                a[j] = x;
            }
    }
    uint32_t s = 0;
    for (int j = 0; j < 8; ++j) s ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int instr_per_elem, uint32_t* out) {
    for (int wg = 1; wg <= 4; wg *= 2) {
        const int iters = 20000;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256 * wg), dim3(256), 0, 0, out, 10, 3u);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 3u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)iters * 64 * instr_per_elem * wg;       // wave-instructions per SIMD (1 wave per SIMD per workgroup)
        printf("%-34s waves/SIMD=%d  %.2f cycles per wave-instruction @2.4GHz\n", name, wg, ms * 1e-3 * 2.4e9 / inst);
    }
}
int main() {
    uint32_t* out; (void)hipMalloc(&out, 1024 * 256 * 4);
    run<0>("v_perm_b32", 1, out);
    run<1>("v_cvt_f32_ubyte1 + v_xor", 2, out);
    run<2>("v_or + v_and + v_sub_f32", 3, out);
    run<3>("v_alignbyte_b32", 1, out);
    run<4>("v_add + v_xor", 2, out);
    return 0;
}
