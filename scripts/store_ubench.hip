// Store-pattern micro-benchmark (gfx950): one persistent 512-thread workgroup per CU writes `per_cu` KB-blocks of a 6.7 GB
// tensor; what differs is how the lanes of a 16-byte store instruction map to addresses.
//   0: whole lines  -- lane l writes bytes [16 l, 16 l + 16) of a 1 KB block (8 full 128-byte lines per instruction)
//   1: transposed-accumulator epilogue -- lane (i = l & 31, h = l >> 5), store g: pixel i (128 B apart), bytes 32 g + 16 h:
//      32 lines x 32 bytes per instruction, 4 instructions complete the 32 lines (a 4 KB tile)
//   2: row-major accumulator epilogue -- dword stores, lanes 0-31 one 128-byte line, lanes 32-63 another (16 per 4 KB tile)
//   3: as 1, but the four stores of a tile are separated by ~2000 cycles of ALU work (spread over a phase)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/store_ubench.bin scripts/store_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long tiles_total, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    float4 v = make_float4(lane, wave, blockIdx.x, 1.f);
    // a tile = 4 KB (32 pixels x 32 channels fp32); tiles are dealt to (workgroup, wave) round-robin in chunks of 13 (one image)
    for (long img = blockIdx.x; img * 13 < tiles_total; img += gridDim.x) {
        for (int t = wave; t < 13; t += 8) {
            float* base = out + (img * 13 + t) * 1024;
            if (MODE == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(base + g * 256 + lane * 4) = v;
            } else if (MODE == 1 || MODE == 3) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<float4*>(base + i * 32 + g * 8 + h * 4) = v;
                    if (MODE == 3) {
                        float a = v.x;
                        for (int s = 0; s < spin; ++s) a = a * 1.0001f + 0.5f;
                        v.x = a;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = v.x;
            }
            if (MODE != 3 && spin) {
                float a = v.x;
                for (int s = 0; s < 4 * spin; ++s) a = a * 1.0001f + 0.5f;
                v.x = a;
            }
        }
    }
}

int main(int argc, char** argv) {
    const long B = 131072, tiles = B * 13;
    float* out;
    hipMalloc(&out, tiles * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, int spin) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, tiles, spin);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s spin %4d: %.3f ms  %.2f TB/s\n", name, spin, ms, tiles * 4096.0 / ms / 1e9);
    };
    for (int spin : {0, 100, 400}) {
        run(k<0>, "0 whole lines, 16 B per lane", spin);
        run(k<1>, "1 transposed epilogue (32 x 32 B per store)", spin);
        run(k<2>, "2 row-major dword stores (2 lines per store)", spin);
        run(k<3>, "3 transposed, stores spread", spin);
    }
    return 0;
}
