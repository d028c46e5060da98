// Convolution layers at latency-bound batch sizes (the DQN learner's batch 32 / 64, actor batches): register-direct implicit GEMMs on the
// fp32 matrix pipe (round 6).
//
// The tiled engine (gemm.hip.h) stages one 128 x 32 operand tile per barrier pair with ONE tile of global loads in flight: at a few dozen
// workgroups every k step costs a full memory round trip (2.7 - 5.5 us per step measured on the Q-network's SAME-padded conv layers:
// 13 - 26 us per forward layer, 22 - 50 us per data gradient).  Here a workgroup owns 32 output rows x 32 TN columns, its NW waves split the
// reduction and keep U chunks of 16-byte (A) / 4-byte (B) loads in flight straight into MFMA operand registers -- no LDS staging, no barrier
// in the main loop --, and the NW partial tiles are summed through LDS in wave order (deterministic).  Hundreds of small workgroups
// instead of dozens of large ones: two or more are co-resident per CU and cover each other's load latency.
// Operand trick as in qheads.hip.h: lane (r, half) of v_mfma_f32_32x32x2_f32 supplies A[row r][kk = half] and B[kk = half][col r]; chunk c
// of 8 reduction indices gives half h the four consecutive indices 8c + 4h .. + 3 (one per MFMA step), so an operand whose reduction index
// is contiguous in memory arrives as one 16-byte load per chunk.
#pragma once

namespace mrl {

// Predicated loads WITHOUT control flow: the address falls back to a location that is always readable and the value is selected
// afterwards.  `ok ? *p : 0` compiles to a branch around the load, and behind a branch the compiler no longer knows how many loads are in
// flight: every wait becomes s_waitcnt vmcnt(0) and the ping-pong prefetch below is serialised (measured: 173 branches, only vmcnt(0)).
__device__ __forceinline__ float4 sel4(bool ok, float4 v) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); }
__device__ __forceinline__ float4 ldz4(const float* p, const float* safe, bool ok) { return sel4(ok, *reinterpret_cast<const float4*>(ok ? p : safe)); }
__device__ __forceinline__ float ldz1(const float* p, const float* safe, bool ok) { const float v = *(ok ? p : safe); return ok ? v : 0.f; }
// 16 bytes from an address that is 16-byte (A16) or only 8-byte aligned (weights inside the flat parameter vector)
template <bool A16> __device__ __forceinline__ float4 ldz4w(const float* p, const float* safe, bool ok) {
    const float* q = ok ? p : safe;
    if (A16) return sel4(ok, *reinterpret_cast<const float4*>(q));
    const float2 lo = *reinterpret_cast<const float2*>(q), hi = *reinterpret_cast<const float2*>(q + 2);
    return sel4(ok, make_float4(lo.x, lo.y, hi.x, hi.y));
}

// ---- forward: out[pixel][n] = act(bias[n] + sum_k patch(pixel)[k] W[k][n]) -----------------------------------------------------------------
// SRC as conv_ld (0: fp32 activations, 1: uint8 pixels / 255, 2: fp32 pixels / 255); SAME / VALID padding through ConvGeom
// Two independent problems of the same layer shape in one launch (the DQN step's online pass of 2 B rows and target pass of B rows: other
// weights, other input, no reason to run one after the other): blocks [0, mt0) belong to p0, the rest to p1.
struct ConvSkinnyProb { ConvGeom g; const float* w; const float* bias; float* out; };
template <int SRC, int TN, int NW>
__global__ __launch_bounds__(64 * NW) void conv_skinny_fwd_kernel(ConvSkinnyProb p0, ConvSkinnyProb p1, int mt0, int NF, int act) {
    __shared__ float red[NW][32][33];
    // the wave index as a SCALAR: chunk numbers, tap decomposition and the weight addresses below are wave-uniform and stay on the
    // scalar unit (747 -> 477 vector instructions in the kernel; -10 % run time)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hf = lane >> 5;
    const bool second = (int)blockIdx.x >= mt0;
    const ConvSkinnyProb& P = second ? p1 : p0;
    const ConvGeom& g = P.g;
    const float* __restrict__ w = P.w;
    const float* __restrict__ bias = P.bias;
    float* __restrict__ out = P.out;
    const int m0 = ((int)blockIdx.x - (second ? mt0 : 0)) * 32, n0 = blockIdx.y * 32 * TN;
    const int kc = g.kconv >> 3, per = (kc + NW - 1) / NW, cb = wave * per, ce = min(kc, cb + per);
    // this lane's pixel
    const int m = min(m0 + r, g.npix - 1);
    const int ohw = g.OH * g.OW;
    const int b = (int)g.d_ohw.div((uint32_t)m), rr = m - b * ohw;
    const int oy = (int)g.d_ow.div((uint32_t)rr), ox = rr - oy * g.OW;
    const long img = g.srow ? (long)g.srow[b] : (long)b;
    const int iy0 = oy * g.stride - g.pad_t, ix0 = ox * g.stride - g.pad_l;
    const long base = ((img * g.H + iy0) * g.W + ix0) * g.C;
    const bool rowlive = m0 + r < g.npix;
    const uint32_t wlane = (uint32_t)(r + 4 * hf * NF);      // this lane's part of a weight address (NF % 32 == 0: every column exists)
    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // chunks in groups of U, the next group's loads issued before the current group's MFMAs (two register sets, ping-pong)
    constexpr int U = TN == 2 ? 2 : 4;
    float4 fa[2][U];
    float fb[2][U][TN][4];
    auto load = [&](int set, int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = c + u < ce;                    // wave-uniform
            const int kb = 8 * min(c + u, kc - 1);           // dead chunks read a valid one, their values are zeroed
            // the two halves' taps, on the scalar unit; the lane picks its own
            const int ky0 = (int)g.d_rowk.div((uint32_t)kb), kr0 = kb - ky0 * g.rowk, kx0 = (int)g.d_c.div((uint32_t)kr0);
            const int ky1 = (int)g.d_rowk.div((uint32_t)(kb + 4)), kr1 = kb + 4 - ky1 * g.rowk, kx1 = (int)g.d_c.div((uint32_t)kr1);
            const long ko0 = (long)ky0 * g.W * g.C + kr0, ko1 = (long)ky1 * g.W * g.C + kr1;
            const int ky = hf ? ky1 : ky0, kx = hf ? kx1 : kx0;
            const bool ok = live && rowlive && (unsigned)(iy0 + ky) < (unsigned)g.H && (unsigned)(ix0 + kx) < (unsigned)g.W;
            fa[set][u] = sel4(ok, conv_ld<SRC>(g.p, ok ? base + (hf ? ko1 : ko0) : 0L));
            const float* wb = w + ((long)kb * NF + n0);      // scalar base + 32-bit lane offset
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = (wb + (j * NF + 32 * t))[wlane];
                    fb[set][u][t][j] = live ? v : 0.f;
                }
        }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].x, fb[set][u][t][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].y, fb[set][u][t][1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].z, fb[set][u][t][2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].w, fb[set][u][t][3], acc[t], 0, 0, 0);
            }
    };
    load(0, cb);                                   // (every load is predicated on its own chunk: loads past `ce` fetch nothing new)
    for (int c = cb; c < ce; c += 2 * U) {
        load(1, c + U);
        mma(0);
        load(0, c + 2 * U);
        if (c + U < ce) mma(1);                    // wave-uniform; no load inside a branch -> exact wait counts
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        if (t) __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][(i >> 2) * 8 + hf * 4 + (i & 3)][r] = acc[t][i];      // row = pixel, column = filter
        __syncthreads();
        for (int idx = threadIdx.x; idx < 1024; idx += 64 * NW) {
            const int p = idx >> 5, n = n0 + 32 * t + (idx & 31);
            if (m0 + p >= g.npix || n >= NF) continue;
            float v = red[0][p][idx & 31];
#pragma unroll
            for (int q = 1; q < NW; ++q) v += red[q][p][idx & 31];
            out[(long)(m0 + p) * NF + n] = act_fwd(v + bias[n], act);
        }
    }
}

template <int SRC>
static hipError_t launch_conv_skinny_fwd(const ConvGeom& g, const float* w, const float* bias, float* out, int NF, int act, hipStream_t st,
                                         const ConvSkinnyProb* second = nullptr) {
    const ConvSkinnyProb p0{g, w, bias, out};
    const ConvSkinnyProb& p1 = second ? *second : p0;
    const int mt0 = (g.npix + 31) / 32, mt = mt0 + (second ? (second->g.npix + 31) / 32 : 0);
    if (NF % 64 == 0) {
        if (g.kconv > 256) hipLaunchKernelGGL((conv_skinny_fwd_kernel<SRC, 2, 8>), dim3(mt, NF / 64), dim3(512), 0, st, p0, p1, mt0, NF, act);
        else hipLaunchKernelGGL((conv_skinny_fwd_kernel<SRC, 2, 4>), dim3(mt, NF / 64), dim3(256), 0, st, p0, p1, mt0, NF, act);
    } else {
        if (g.kconv > 256) hipLaunchKernelGGL((conv_skinny_fwd_kernel<SRC, 1, 8>), dim3(mt, (NF + 31) / 32), dim3(512), 0, st, p0, p1, mt0, NF, act);
        else hipLaunchKernelGGL((conv_skinny_fwd_kernel<SRC, 1, 4>), dim3(mt, (NF + 31) / 32), dim3(256), 0, st, p0, p1, mt0, NF, act);
    }
    return hipGetLastError();
}

// ---- data gradient (gather form, stride-parity classes as DgradGeom): rows = input pixels of class z, k = (tap, filter) -----------------------
// dX[b, iy, ix, c] = act'(h) * sum_{tap (a, b2), n} dz[b, yy - a, xx - b2, n] W[py + s a, px + s b2, c, n];  NF % 8 == 0: a chunk of 8 k
// stays inside one tap, so both operands arrive as 16-byte loads along n.  grid (row tiles of 32, channel tiles of 32 TN, classes)
template <int TN, int NW, bool W16>
__global__ __launch_bounds__(64 * NW) void conv_skinny_dgrad_kernel(DgradGeom g, const float* __restrict__ dz, const float* __restrict__ w,
                                                                    EpiDgradConv ef) {
    __shared__ float red[NW][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hf = lane >> 5;
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32 * TN, z = blockIdx.z;
    const int py = z / g.stride, px = z - py * g.stride;
    const int perimg = g.HY * g.WX, Mz = g.B * perimg;
    const int kc = (g.taps * g.taps * g.NF) >> 3, per = (kc + NW - 1) / NW, cb = wave * per, ce = min(kc, cb + per);
    const int m = min(m0 + r, Mz - 1);
    const bool rowlive = m0 + r < Mz;
    const int b = (int)g.d_per.div((uint32_t)m), rr = m - b * perimg;
    const int yy = (int)g.d_wx.div((uint32_t)rr), xx = rr - yy * g.WX;
    const long pb = (long)b * g.OH * g.OW;
    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    constexpr int U = TN == 2 ? 2 : 4;
    float4 fa[2][U], fb[2][U][TN];
    auto load = [&](int set, int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = c + u < ce;
            const int k = 8 * (c + u) + 4 * hf;
            const int tap = (int)g.d_nf.div((uint32_t)k), n = k - tap * g.NF;
            const int a = (int)g.d_taps.div((uint32_t)tap), b2 = tap - a * g.taps;
            const int oy = yy - a, ox = xx - b2;
            const bool oka = live && rowlive && (unsigned)oy < (unsigned)g.OH && (unsigned)ox < (unsigned)g.OW;
            fa[set][u] = ldz4(dz + (pb + oy * g.OW + ox) * g.NF + n, dz, oka);
            const int ky = py + g.stride * a, kx = px + g.stride * b2;
            const bool okb = live && ky < g.rf && kx < g.rf;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int cc = c0 + 32 * t + r;
                fb[set][u][t] = ldz4w<W16>(w + ((long)(ky * g.rf + kx) * g.C + cc) * g.NF + n, w, okb && cc < g.C);
            }
        }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].x, fb[set][u][t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].y, fb[set][u][t].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].z, fb[set][u][t].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].w, fb[set][u][t].w, acc[t], 0, 0, 0);
            }
    };
    load(0, cb);                                   // (every load is predicated on its own chunk: loads past `ce` fetch nothing new)
    for (int c = cb; c < ce; c += 2 * U) {
        load(1, c + U);
        mma(0);
        load(0, c + 2 * U);
        if (c + U < ce) mma(1);                    // wave-uniform; no load inside a branch -> exact wait counts
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        if (t) __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][(i >> 2) * 8 + hf * 4 + (i & 3)][r] = acc[t][i];      // row = input pixel, column = channel
        __syncthreads();
        for (int idx = threadIdx.x; idx < 1024; idx += 64 * NW) {
            const int p = idx >> 5, cc = c0 + 32 * t + (idx & 31);
            if (m0 + p >= Mz || cc >= g.C) continue;
            const long o = ef.addr(m0 + p, cc, z);
            if (o < 0) continue;
            float v = red[0][p][idx & 31];
#pragma unroll
            for (int q = 1; q < NW; ++q) v += red[q][p][idx & 31];
            ef.put(o, v, ef.aux(o, cc));
        }
    }
}

template <bool W16>
static hipError_t launch_conv_skinny_dgrad_a(const DgradGeom& g, const float* dz, const float* w, const EpiDgradConv& ef, hipStream_t st) {
    const int Mz = g.B * g.HY * g.WX, mt = (Mz + 31) / 32, zc = g.stride * g.stride;
    const bool deep = g.taps * g.taps * g.NF > 256;
    if (g.C % 64 == 0) {
        if (deep) hipLaunchKernelGGL((conv_skinny_dgrad_kernel<2, 8, W16>), dim3(mt, g.C / 64, zc), dim3(512), 0, st, g, dz, w, ef);
        else hipLaunchKernelGGL((conv_skinny_dgrad_kernel<2, 4, W16>), dim3(mt, g.C / 64, zc), dim3(256), 0, st, g, dz, w, ef);
    } else {
        if (deep) hipLaunchKernelGGL((conv_skinny_dgrad_kernel<1, 8, W16>), dim3(mt, (g.C + 31) / 32, zc), dim3(512), 0, st, g, dz, w, ef);
        else hipLaunchKernelGGL((conv_skinny_dgrad_kernel<1, 4, W16>), dim3(mt, (g.C + 31) / 32, zc), dim3(256), 0, st, g, dz, w, ef);
    }
    return hipGetLastError();
}
static hipError_t launch_conv_skinny_dgrad(const DgradGeom& g, const float* dz, const float* w, const EpiDgradConv& ef, hipStream_t st) {
    return (uintptr_t)w % 16 == 0 ? launch_conv_skinny_dgrad_a<true>(g, dz, w, ef, st) : launch_conv_skinny_dgrad_a<false>(g, dz, w, ef, st);
}

}  // namespace mrl
