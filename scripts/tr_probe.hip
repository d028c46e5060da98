// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read): which lane receives which source element, and what the
// access costs for the per-lane address patterns of the image-resident weight-gradient kernel (wgradtr.hip.h).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/tr_probe.bin scripts/tr_probe.hip && scripts/tr_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;

__device__ __forceinline__ v4i16 tr_read(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(p));
}

// semantics: LDS element e holds the value e; lane l reads at element offset off[l]; out[l*4+j] = what it got
__global__ void sem_kernel(const int* off, int* out) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    for (int e = threadIdx.x; e < 16384; e += blockDim.x) lds[e] = (uint16_t)e;
    __syncthreads();
    const int l = threadIdx.x;
    v4i16 r = tr_read(lds + off[l]);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)r[j];
}

// timing: every wave issues NREAD tr reads (or plain b64 / b128 reads) per iteration with per-lane offsets off[l] + immediates
template <int MODE>
__global__ __launch_bounds__(512) void time_kernel(const int* off, long long* cyc, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    for (int e = threadIdx.x; e < 65536; e += blockDim.x) lds[e] = (uint16_t)e;
    __syncthreads();
    const int l = threadIdx.x & 63;
    const uint16_t* p = lds + off[l] + (threadIdx.x >> 6) * 64;
    int acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MODE == 0) {
                v4i16 r = tr_read(p + q * 2048);
                acc += r[0] + r[1] + r[2] + r[3];
            } else if (MODE == 1) {
                uint2 r = *reinterpret_cast<const uint2*>(p + q * 2048);
                acc += r.x + r.y;
            } else {
                uint4 r = *reinterpret_cast<const uint4*>(p + q * 2048);
                acc += r.x + r.y + r.z + r.w;
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x7fffffff) sink[0] = acc;
}

static void run_time(const char* name, const std::vector<int>& off, int mode, int waves) {
    int* d_off; long long* d_cyc; int* d_sink;
    hipMalloc(&d_off, 64 * 4); hipMalloc(&d_cyc, 8 * 8); hipMalloc(&d_sink, 4);
    hipMemcpy(d_off, off.data(), 64 * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(time_kernel<0>, dim3(1), dim3(64 * waves), 131072, 0, d_off, d_cyc, iters, d_sink);
        else if (mode == 1) hipLaunchKernelGGL(time_kernel<1>, dim3(1), dim3(64 * waves), 131072, 0, d_off, d_cyc, iters, d_sink);
        else hipLaunchKernelGGL(time_kernel<2>, dim3(1), dim3(64 * waves), 131072, 0, d_off, d_cyc, iters, d_sink);
        hipDeviceSynchronize();
    }
    long long c[8];
    hipMemcpy(c, d_cyc, 64, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
    printf("  %-44s mode %d waves %d: %.2f cycles per wave-read, %.2f per CU-read\n", name, mode, waves,
           (double)mx / (iters * 16.0), (double)mx / (iters * 16.0 * waves));
    hipFree(d_off); hipFree(d_cyc); hipFree(d_sink);
}

int main() {
    hipFuncSetAttribute((const void*)time_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)time_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)time_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // ---- semantics
    int* d_off; int* d_out;
    hipMalloc(&d_off, 64 * 4); hipMalloc(&d_out, 256 * 4);
    for (int pat = 0; pat < 3; ++pat) {
        std::vector<int> off(64);
        for (int l = 0; l < 64; ++l)
            off[l] = pat == 0 ? 4 * l : pat == 1 ? 4 * ((l * 37) % 64) : 64 * (l & 15) + 4 * (l >> 4);   // elements (8-byte aligned)
        hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 32768, 0, d_off, d_out);
        std::vector<int> out(256);
        hipMemcpy(out.data(), d_out, 1024, hipMemcpyDeviceToHost);
        printf("pattern %d (lane l reads 4 elements at element offset off[l]); got[lane][j] as (source lane, element):\n", pat);
        // map value -> (source lane, element index) through the offsets
        bool modelA = true;
        for (int l = 0; l < 64; ++l) {
            printf("  l%02d:", l);
            for (int j = 0; j < 4; ++j) {
                int v = out[l * 4 + j], sl = -1, se = -1;
                for (int s = 0; s < 64; ++s)
                    if (v >= off[s] && v < off[s] + 4) { sl = s; se = v - off[s]; }
                printf(" (%2d,%d)", sl, se);
                // model A: R_l[j] = D_{16*(l>>4) + 4*j + ((l&15)>>2)}[l&3]
                if (sl != 16 * (l >> 4) + 4 * j + ((l & 15) >> 2) || se != (l & 3)) modelA = false;
            }
            if ((l & 3) == 3) printf("\n");
        }
        printf("  model A (R_l[j] = D_{16(l>>4)+4j+((l&15)>>2)}[l&3]) holds: %s\n", modelA ? "YES" : "NO");
    }
    // ---- timing of candidate address patterns.  Lane roles for an MFMA 32x32x16 operand fragment read with two tr reads:
    //      g = l>>4: mb = g&1 (16-row half of the 32 rows), h = g>>1 (k half); p = l&15: k row = p>>2, column quad = p&3
    auto pattern = [&](int pix_stride_el, int chan0_el, auto pixel_of) {
        std::vector<int> off(64);
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, mb = g & 1, h = g >> 1, p = l & 15, kr = p >> 2, cq = p & 3;
            off[l] = pixel_of(8 * h + kr) * pix_stride_el + chan0_el + 16 * mb + 4 * cq;
        }
        return off;
    };
    for (int waves : {1, 4, 8}) {
        run_time("dense 4x16 blocks (ideal)", [&] { std::vector<int> o(64); for (int l = 0; l < 64; ++l) o[l] = 4 * l; return o; }(), 0, waves);
        run_time("plain ds_read_b64 dense", [&] { std::vector<int> o(64); for (int l = 0; l < 64; ++l) o[l] = 4 * l; return o; }(), 1, waves);
        run_time("plain ds_read_b128 dense", [&] { std::vector<int> o(64); for (int l = 0; l < 64; ++l) o[l] = 8 * l; return o; }(), 2, waves);
        run_time("C=32 px 64B, consecutive pixels", pattern(32, 0, [](int k) { return k; }), 0, waves);
        run_time("C=32 px 64B, every 2nd pixel", pattern(32, 0, [](int k) { return 2 * k; }), 0, waves);
        run_time("C=32 px 80B (pad), every 2nd pixel", pattern(40, 0, [](int k) { return 2 * k; }), 0, waves);
        run_time("C=32 px 96B (pad), every 2nd pixel", pattern(48, 0, [](int k) { return 2 * k; }), 0, waves);
        run_time("C=64 px 128B, consecutive pixels", pattern(64, 0, [](int k) { return k; }), 0, waves);
        run_time("C=64 px 128B, consecutive, 2nd half", pattern(64, 32, [](int k) { return k; }), 0, waves);
        run_time("C=64 px 144B (pad 16B), consecutive", pattern(72, 0, [](int k) { return k; }), 0, waves);
        run_time("C=64 px 160B (pad 32B), consecutive", pattern(80, 0, [](int k) { return k; }), 0, waves);
        run_time("C=64 px 192B (pad 64B), consecutive", pattern(96, 0, [](int k) { return k; }), 0, waves);
        run_time("C=64 px 136B (pad 8B), consecutive", pattern(68, 0, [](int k) { return k; }), 0, waves);
        run_time("C=32 px 72B (pad 8B), every 2nd pixel", pattern(36, 0, [](int k) { return 2 * k; }), 0, waves);
        run_time("C=32 px 72B (pad 8B), consecutive", pattern(36, 0, [](int k) { return k; }), 0, waves);
    }
    return 0;
}
