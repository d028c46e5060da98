// "Weights-resident" fp32-MFMA convolution GEMM for gfx950 (MI355X): the second GEMM engine of
// libmrl, used where the WHOLE B operand (a conv layer's filter bank: 32-148 KB) fits the CU's
// 160 KB LDS -- NatureCNN conv forward (a2c/utils.py:37-56 via common/models.py:15-26) and conv
// data-gradient.
//
// Why it exists (measured on MI355X, profiles/): with v_mfma_f32_32x32x2_f32 at 64 cycles per
// instruction the tiled kernel (gemm.hip.h) is bound by what happens BETWEEN MFMAs -- barriers,
// LDS round trips of the A tile, staging VALU -- and by occupancy, not by any bandwidth.  For an
// operand whose fragment is consumed by exactly one wave the LDS round trip is a pure pass-through:
// the MFMA A fragment of lane (i, h) is 4 consecutive k of row i, i.e. one 16-byte global load.  So:
//   * the filter bank is transposed into LDS ONCE per workgroup (Wt[n][K+4], conflict-free
//     ds_read_b128 fragments) and stays there for the kernel's lifetime;
//   * each wave owns whole 32-row output tiles and streams its A fragments global -> VGPR directly
//     (im2col / gather / u8->f32 addressing done per lane, PF blocks in flight), no LDS, NO BARRIER
//     in the main loop; waves are fully independent, 8-16 of them per CU hide each other's latency;
//   * one persistent workgroup per CU (grid = #CUs), tiles handed out round-robin; the last group of
//     a tile already prefetches the first group of the wave's next tile (no per-tile pipeline bubble).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm.hip.h"

namespace mrl {

// ------------------------------------------------------------------------------------------
// A-fragment loaders.  A row's K dimension is consumed in GROUPS of PF blocks; a block is 2*KL k
// (lane half h holds KL consecutive k).  Interface:
//   static constexpr int KL;
//   struct RowState { int z; ... };     per (wave, tile): class + row addressing + group cursor
//   struct Frag;                        raw registers of one block
//   row_init(rs, tile, i, h)            tile -> class z, row (clamped), cursor at group 0
//   prep_group<PF>(rs)                  fixes the base address of the next group, advances the cursor
//   load_one<PF>(rs, u) -> Frag         block u of the prepared group (constant offsets from its base)
//   convert(Frag, float (&a)[KL])
// Everything in prep_group / load_one is straight-line code: the software pipeline of the kernel
// must not contain control flow or the compiler serialises the loads (s_waitcnt vmcnt(0) per block).
// ------------------------------------------------------------------------------------------

// conv forward: rows = output pixels m = (b, oy, ox) (optionally image-gathered through srow),
// k = (ky, kx, c) in HWIO order.
//   U8 : pixels are uint8 scaled by 1/255 (models.py:19); a lane holds 16 consecutive k (16 bytes);
//        requires rf*C == 32 (one block per patch row) and rf % PF == 0 -- group = PF patch rows.
//   F32: a lane holds 4 consecutive k; requires (rf*C) % (8*PF) == 0 -- group = 8*PF k of one patch row.
template <bool U8>
struct WresFwdA : ConvGeom {
    static constexpr int KL = U8 ? 16 : 4;
    struct RowState { int z; long base; long koff; int kr; const uint8_t* g; };
    struct Frag { uint4 u; };
    __device__ __forceinline__ void row_init(RowState& s, long tile, int i, int h) const {
        s.z = 0;
        int m = (int)min(tile * 32 + i, (long)npix - 1);
        const int ohw = OH * OW;
        int b = (int)d_ohw.div((uint32_t)m), r = m - b * ohw;
        int oy = (int)d_ow.div((uint32_t)r), ox = r - oy * OW;
        long img = srow ? (long)srow[b] : (long)b;
        s.base = ((img * H + oy * stride) * W + ox * stride) * C + h * KL;
        s.koff = 0;
        s.kr = 0;
    }
    template <int PF> __device__ __forceinline__ void prep_group(RowState& s) const {
        if constexpr (U8) {
            s.g = static_cast<const uint8_t*>(p) + s.base + s.koff;
            s.koff += (long)PF * W * C;
        } else {
            s.g = reinterpret_cast<const uint8_t*>(static_cast<const float*>(p) + s.base + s.koff + s.kr);
            s.kr += 8 * PF;
            if (s.kr >= rowk) { s.kr = 0; s.koff += (long)W * C; }
        }
    }
    template <int PF> __device__ __forceinline__ Frag load_one(const RowState& s, int u) const {
        Frag f;
        if constexpr (U8) f.u = *reinterpret_cast<const uint4*>(s.g + (long)u * W * C);
        else f.u = *reinterpret_cast<const uint4*>(s.g + u * 32);
        return f;
    }
    __device__ __forceinline__ void convert(const Frag& f, float (&a)[KL]) const {
        if constexpr (U8) {
            const uint32_t w[4] = {f.u.x, f.u.y, f.u.z, f.u.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q * 4 + 0] = u8_over_255((float)(w[q] & 0xff));
                a[q * 4 + 1] = u8_over_255((float)((w[q] >> 8) & 0xff));
                a[q * 4 + 2] = u8_over_255((float)((w[q] >> 16) & 0xff));
                a[q * 4 + 3] = u8_over_255((float)(w[q] >> 24));
            }
        } else {
            a[0] = __uint_as_float(f.u.x); a[1] = __uint_as_float(f.u.y);
            a[2] = __uint_as_float(f.u.z); a[3] = __uint_as_float(f.u.w);
        }
    }
};

// conv data-gradient (gather form, see DgradGeom in gemm.hip.h): tile -> (parity class z, 32 rows
// (b, yy, xx) of that class); k = (tap (a, b2), n); a lane holds 4 consecutive n of one tap.
// Requires NF == 8*PF: one group = one tap.  Taps that fall outside the output map read `zeros`
// (>= NF floats of 0.0f) so the load stays unconditional.
struct WresDgradA : DgradGeom {
    static constexpr int KL = 4;
    const float* dz;             // [B, OH, OW, NF]
    const float* zeros;
    long tiles_per_class;
    struct RowState { int z; long pbo; int yy, xx; int a, b2; const float* g; const float* zq; };
    struct Frag { float4 v; };
    __device__ __forceinline__ void row_init(RowState& s, long tile, int i, int h) const {
        s.z = (int)(tile / tiles_per_class);
        long m = (tile - (long)s.z * tiles_per_class) * 32 + i;
        const int per = HY * WX;
        int pb = 0;
        if (m < (long)B * per) {
            int b = (int)d_per.div((uint32_t)m), r = (int)m - b * per;
            s.yy = (int)d_wx.div((uint32_t)r);
            s.xx = r - s.yy * WX;
            pb = b * OH * OW;
        } else {
            s.yy = -0x10000; s.xx = 0;                 // never in range -> zeros
        }
        s.a = 0; s.b2 = 0;
        s.zq = zeros + h * 4;
        s.g = s.zq;
        s.pbo = (long)pb * NF + h * 4;                 // element offset of (b, 0, 0, n = 4h)
    }
    template <int PF> __device__ __forceinline__ void prep_group(RowState& s) const {
        const int oy = s.yy - s.a, ox = s.xx - s.b2;
        const bool ok = (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW;
        const float* q = dz + (s.pbo + (long)(oy * OW + ox) * NF);
        s.g = ok ? q : s.zq;
        if (++s.b2 == taps) { s.b2 = 0; ++s.a; }
    }
    template <int PF> __device__ __forceinline__ Frag load_one(const RowState& s, int u) const {
        Frag f;
        f.v = *reinterpret_cast<const float4*>(s.g + u * 8);
        return f;
    }
    __device__ __forceinline__ void convert(const Frag& f, float (&a)[KL]) const {
        a[0] = f.v.x; a[1] = f.v.y; a[2] = f.v.z; a[3] = f.v.w;
    }
};

// ------------------------------------------------------------------------------------------
// B (filter bank) descriptions: element (z, k, n) of the GEMM's B operand, read from HBM once per WG
// ------------------------------------------------------------------------------------------
struct WresFwdB {            // W is HWIO flattened [K][N]
    const float* w; int N;
    __device__ __forceinline__ float get(int, int k, int n) const { return w[(long)k * N + n]; }
};
struct WresDgradB : DgradGeom {   // class z = (py, px); k = (tap, n'); column = input channel c
    const float* w;
    __device__ __forceinline__ float get(int z, int k, int c) const {
        int py = z / stride, px = z - py * stride;
        int tap = k / NF, n = k - tap * NF;
        int a = tap / taps, b2 = tap - a * taps;
        int ky = py + stride * a, kx = px + stride * b2;
        if (ky >= rf || kx >= rf) return 0.f;
        return w[((long)(ky * rf + kx) * C + c) * NF + n];
    }
};

// ------------------------------------------------------------------------------------------
// Epilogues (three phases like gemm.hip.h: all auxiliary loads of a tile are issued before its stores):
//   long addr(tile, z, row_in_tile, col) (-1: no destination);  float aux(o, col);  void put(o, acc, aux)
// ------------------------------------------------------------------------------------------
struct WresEpiBiasAct {      // out[m*ld + n] = act(acc + bias[n]),  m = tile*32 + row
    float* out; long ld; const float* bias; int act; long rows; int ncols;
    // optional ReLU bit mask of the output (bit e of word e/32 <=> out[e] > 0): one word per 32 consecutive columns,
    // written with one ballot per row pair -- 1 bit instead of 32 for the data-gradient kernel that applies act'
    uint32_t* mask = nullptr;
    __device__ __forceinline__ long addr(long tile, int, int row, int col) const {
        long m = tile * 32 + row;
        return (m < rows && col < ncols) ? m * ld + col : -1;
    }
    __device__ __forceinline__ float aux(long, int col) const { return bias[min(col, ncols - 1)]; }
    __device__ __forceinline__ void put(long o, float acc, float b) const { out[o] = act_fwd(acc + b, act); }
};
struct WresEpiDgrad : DgradGeom {   // scatter class rows back to NHWC, masked by act'(h_prev)
    float* out; const float* hprev; int act; long tiles_per_class;
    __device__ __forceinline__ long addr(long tile, int z, int row, int col) const {
        long m = (tile - (long)z * tiles_per_class) * 32 + row;
        const int per = HY * WX;
        if (m >= (long)B * per || col >= C) return -1;
        int b = (int)d_per.div((uint32_t)m), r = (int)m - b * per;
        int yy = (int)d_wx.div((uint32_t)r), xx = r - yy * WX;
        int py = z / stride, px = z - py * stride;
        int iy = yy * stride + py, ix = xx * stride + px;
        if (iy >= H || ix >= W) return -1;
        return ((long)(b * H + iy) * W + ix) * C + col;
    }
    __device__ __forceinline__ float aux(long o, int) const { return hprev[o]; }
    __device__ __forceinline__ void put(long o, float acc, float hv) const { out[o] = acc * act_bwd_from_out(hv, act); }
};

// ------------------------------------------------------------------------------------------
template <class AL, class BL, class EF, int NT, int PF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void wres_kernel(AL al, BL bl, EF ef, int zc, int K, int N, int KP,
                                                          long total_tiles) {
    constexpr int KL = AL::KL;
    extern __shared__ __attribute__((aligned(16))) float wt[];   // [zc][NT*32][KP]
    const int tid = threadIdx.x;
    // ---- one-time transpose of the filter bank into LDS (columns >= N and the k pad are zeroed)
    {
        const int rowsz = NT * 32;
        const int tot = zc * rowsz * KP;
        for (int e = tid; e < tot; e += WAVES * 64) wt[e] = 0.f;
        __syncthreads();
        const int nel = zc * K * N;
        for (int e = tid; e < nel; e += WAVES * 64) {
            int t = e / N, n = e - t * N;
            int z = t / K, k = t - z * K;
            wt[(z * rowsz + n) * KP + k] = bl.get(z, k, n);
        }
        __syncthreads();
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int NG = K / (2 * KL * PF);            // groups of PF blocks per row (host guarantees exact)
    const long stride_t = (long)gridDim.x * WAVES;
    long tile = (long)blockIdx.x * WAVES + wave;
    if (tile >= total_tiles) return;

    typename AL::RowState rs;
    typename AL::Frag fr[PF];
    al.row_init(rs, tile, i, h);
    al.template prep_group<PF>(rs);
#pragma unroll
    for (int u = 0; u < PF; ++u) fr[u] = al.template load_one<PF>(rs, u);
    int zcur = rs.z;

    // one group of PF blocks.  The NEXT group's PF loads are issued first, into a second register
    // set, and fenced with sched_barrier so the compiler cannot sink them behind the MFMAs (it does,
    // which exposes the full memory latency once per group); they have the whole group -- PF*KL*NT
    // MFMAs = 2-4k cycles -- to land.
    auto body = [&](const typename AL::RowState& src, const float* wb, f32x16 (&acc)[NT]) {
        typename AL::Frag fn[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) fn[u] = al.template load_one<PF>(src, u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            float a[KL];
            al.convert(fr[u], a);
#pragma unroll
            for (int q = 0; q < KL / 4; ++q) {
                float4 bq[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bq[nt] = *reinterpret_cast<const float4*>(wb + (long)nt * 32 * KP + u * 2 * KL + q * 4);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q * 4 + 0], bq[nt].x, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q * 4 + 1], bq[nt].y, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q * 4 + 2], bq[nt].z, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q * 4 + 3], bq[nt].w, acc[nt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PF; ++u) fr[u] = fn[u];
    };

    while (true) {
        const float* wz = wt + ((long)zcur * NT * 32 + i) * KP + h * KL;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        for (int g = 0; g + 1 < NG; ++g) {       // steady state: refill from the same row's next group
            al.template prep_group<PF>(rs);
            body(rs, wz + g * (2 * KL * PF), acc);
        }
        // last group of this tile: refill from the NEXT tile's first group (clamped to a valid tile so
        // the loads stay unconditional; the surplus of the final tile is simply never consumed)
        const long next = tile + stride_t;
        typename AL::RowState rn;
        al.row_init(rn, min(next, total_tiles - 1), i, h);
        al.template prep_group<PF>(rn);
        body(rn, wz + (NG - 1) * (2 * KL * PF), acc);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            long o[16];
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = ef.addr(tile, zcur, (r & 3) + 8 * (r >> 2) + 4 * h, nt * 32 + i);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = ef.aux(o[r] < 0 ? 0 : o[r], nt * 32 + i);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (o[r] >= 0) ef.put(o[r], acc[nt][r], x[r]);
        }
        if (next >= total_tiles) break;
        rs = rn;
        zcur = rn.z;
        tile = next;
    }
}

// ------------------------------------------------------------------------------------------
// u8 forward on the BF16 matrix pipe ("bf16 x 3"): the first conv layer's A operand is raw uint8 pixels,
// which are EXACT in bf16 (8 significant bits).  Splitting the fp32 filter (pre-scaled by 1/255) into three
// bf16 planes  w = hi + mid + lo  (24 significant bits, exact) gives  x*w = x*hi + x*mid + x*lo  with every
// product exact in the fp32 accumulator -- fp32-class accuracy (not the bitwise fmaf chain of the fp32 MFMA:
// it differs from (x/255)*w by the rounding of the scale, ~6e-8 relative) at 3 v_mfma_f32_32x32x16_bf16
// (32 cycles each, 16 k) instead of 8 v_mfma_f32_32x32x2_f32 (64 cycles each): 5.3x less matrix time.
// Same structure as wres_kernel: filter planes resident in LDS, A fragments global -> VGPR, no barriers.
// Fragment layout of 32x32x16 bf16: lane (i = l&31, g = l>>5) holds k = 8g .. 8g+7 of its row.  A lane of
// this kernel holds 16 bytes (k = 16h .. 16h+15 of a 32-byte patch row): MFMA block b (0/1) uses its bytes
// 8b .. 8b+7, i.e. the k set {16g + 8b + e}; the B fragment is read from the same k.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct U32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t bf16_rn_bits(float f) {          // round-to-nearest-even, finite inputs
    uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// 4 packed u8 -> 4 bf16 (exact): two dwords of packed pairs
__device__ __forceinline__ void u8x4_to_bf16(uint32_t w, uint32_t& lo, uint32_t& hi) {
    const uint32_t f0 = __float_as_uint((float)(w & 0xff)), f1 = __float_as_uint((float)((w >> 8) & 0xff));
    const uint32_t f2 = __float_as_uint((float)((w >> 16) & 0xff)), f3 = __float_as_uint((float)(w >> 24));
    lo = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
    hi = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
}

template <class EF, int PF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void wres_u8x3_kernel(WresFwdA<true> al, const float* __restrict__ w, EF ef,
                                                               int K, int N, long total_tiles) {
    constexpr int KL = 16;
    extern __shared__ __attribute__((aligned(16))) uint16_t wp[];     // [3][32][KP] bf16 planes, KP = K + 8
    const int KP = K + 8;
    const int tid = threadIdx.x;
    for (int e = tid; e < 3 * 32 * KP; e += WAVES * 64) wp[e] = 0;
    __syncthreads();
    for (int e = tid; e < K * N; e += WAVES * 64) {
        const int k = e / N, n = e - k * N;
        const float v = w[e] / 255.f;                                  // models.py:19 scale folded into the filter
        const uint32_t h0 = bf16_rn_bits(v);
        const float r1 = v - __uint_as_float(h0 << 16);
        const uint32_t h1 = bf16_rn_bits(r1);
        const float r2 = r1 - __uint_as_float(h1 << 16);
        const uint32_t h2 = bf16_rn_bits(r2);
        wp[(0 * 32 + n) * KP + k] = (uint16_t)h0;
        wp[(1 * 32 + n) * KP + k] = (uint16_t)h1;
        wp[(2 * 32 + n) * KP + k] = (uint16_t)h2;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const long stride_t = (long)gridDim.x * WAVES;
    long tile = (long)blockIdx.x * WAVES + wave;
    if (tile >= total_tiles) return;
    const int NG = K / (2 * KL * PF);

    typename WresFwdA<true>::RowState rs;
    typename WresFwdA<true>::Frag fr[PF];
    al.row_init(rs, tile, i, h);
    al.template prep_group<PF>(rs);
#pragma unroll
    for (int u = 0; u < PF; ++u) fr[u] = al.template load_one<PF>(rs, u);
    const uint16_t* wrow = wp + (long)i * KP + 16 * h;                 // + plane*32*KP + block*32 + 8b

    auto body = [&](const typename WresFwdA<true>::RowState& src, int blk0, f32x16& acc) {
        typename WresFwdA<true>::Frag fn[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) fn[u] = al.template load_one<PF>(src, u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            U32x4 a0, a1;                                              // bytes 0-7 / 8-15 of the lane as bf16 x 8
            u8x4_to_bf16(fr[u].u.x, a0.x, a0.y);
            u8x4_to_bf16(fr[u].u.y, a0.z, a0.w);
            u8x4_to_bf16(fr[u].u.z, a1.x, a1.y);
            u8x4_to_bf16(fr[u].u.w, a1.z, a1.w);
            const bf16x8 av0 = __builtin_bit_cast(bf16x8, a0), av1 = __builtin_bit_cast(bf16x8, a1);
            const uint16_t* wb = wrow + (blk0 + u) * 32;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(wb + (long)pl * 32 * KP);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wb + (long)pl * 32 * KP + 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av1, b1, acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PF; ++u) fr[u] = fn[u];
    };

    while (true) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int g = 0; g + 1 < NG; ++g) {
            al.template prep_group<PF>(rs);
            body(rs, g * PF, acc);
        }
        const long next = tile + stride_t;
        typename WresFwdA<true>::RowState rn;
        al.row_init(rn, min(next, total_tiles - 1), i, h);
        al.template prep_group<PF>(rn);
        body(rn, (NG - 1) * PF, acc);
        long o[16];
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = ef.addr(tile, 0, (r & 3) + 8 * (r >> 2) + 4 * h, i);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = ef.aux(o[r] < 0 ? 0 : o[r], i);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (o[r] >= 0) ef.put(o[r], acc[r], x[r]);
        if (ef.mask) {             // lanes 0-31 hold the 32 columns of one output row, lanes 32-63 those of another:
            // a wave ballot per accumulator register = the mask words of two rows; lane L < 32 collects row L's word and the
            // 32 words of the tile leave with one coalesced store (ncols == 32: one word per output row)
            const int lane = (h << 5) | i;
            const int mr = (lane & 3) + 4 * (lane >> 3), mh = (lane >> 2) & 1;
            uint32_t mword = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned long long bal = __ballot(o[r] >= 0 && act_fwd(acc[r] + x[r], ef.act) > 0.f);
                if (lane < 32 && mr == r) mword = (uint32_t)(mh ? (bal >> 32) : bal);
            }
            const long m = tile * 32 + lane;
            if (lane < 32 && m < ef.rows) ef.mask[m] = mword;
        }
        if (next >= total_tiles) break;
        rs = rn;
        tile = next;
    }
}

template <class EF, int PF, int WAVES>
inline hipError_t launch_wres_u8x3(const WresFwdA<true>& al, const float* w, const EF& ef, int K, int N, long total_tiles,
                                   int num_cus, hipStream_t stream) {
    const size_t lds = (size_t)3 * 32 * (K + 8) * sizeof(uint16_t);
    auto kern = wres_u8x3_kernel<EF, PF, WAVES>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    long want = (total_tiles + WAVES - 1) / WAVES;
    int grid = (int)std::min<long>(std::max<long>(want, 1), num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, al, w, ef, K, N, total_tiles);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Exact 3-way bf16 split of fp32 values (the split engines: gemmx6 / dgradx6 / wgradtr / wgradx8): x = x0 + x1 + x2, every
// plane a bf16 value, the sum EXACT.  Each level rounds to NEAREST (ties away from zero: half a unit of the bf16 last place
// added to the magnitude bits, then truncated), so the residual after a plane is at most half a unit of that plane's last
// place: |x1| <= 2^-8 |x|, |x2| <= 2^-17 |x|, and the last residual has at most 8 significant bits -- it IS a bf16 value.
// (Finite inputs below 2^127 (1 - 2^-9): nearer to the largest fp32 the first rounding would carry into the exponent of
// infinity; activations and gradients never get there.)
//
// A product of two fp32 values is the sum of the 9 plane products, each exact in the fp32 accumulator (8 x 8 significant
// bits).  The engines keep the SIX products x_i w_j with i + j <= 2; what they drop is x1 w2 + x2 w1 + x2 w2
// <= (2^-25 + 2^-25 + 2^-34) |x w| -- at most 2^-24 of the product, the rounding of ONE IEEE fp32 multiply, and 3.5e-9 of
// it on average, without bias (an fp32 multiply: 2.1e-8; tests/test_split_arithmetic.py).  A build with -DMRL_PRODUCTS8
// keeps x1 w2 and x2 w1 as well (dropped part < 2^-33 of a product) at 8/6 of the matrix-pipe time.
//
// Cost: 15 full-rate VALU instructions per two values (a truncating split is 11, but its one-sided residuals are twice as
// large: six products would then be off by up to 2^-21 and biased, eight were needed -- 158 ms per benched epoch against 138
// with this split).  Measured and dropped: the hardware conversion v_cvt_pk_bf16_f32 (9 instructions per pair, but the
// A-staging engines ran 35 % SLOWER: the conversion is not a full-rate instruction) and Veltkamp's splitting in packed fp32
// arithmetic (11 v_pk_mul/add_f32 per pair: 147 ms, the packed fp32 operations issue at half rate), and residuals taken
// straight from the packed plane with v_dot2c_f32_bf16 (11 per pair: 147 ms as well, and not exact -- it fails the
// engine-agreement test).
// (A weights-resident forward built on per-lane global -> VGPR fragments was measured TA/L1-bound and is not kept:
// profiles/README.md.)
// ------------------------------------------------------------------------------------------
#ifdef MRL_PRODUCTS8
constexpr bool kCross21 = true;
#else
constexpr bool kCross21 = false;
#endif
constexpr int kSplitProducts = kCross21 ? 8 : 6;
// two floats -> three packed bf16 pairs (plane 0 / 1 / 2; x0 in the low half)
__device__ __forceinline__ void split2_bf16x3(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
#ifdef MRL_FAKE_SPLIT      // timing experiment only (wrong results): 3 instead of 15 VALU instructions per pair
    p0 = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
    p1 = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x05040100u);
    p2 = p0 ^ p1;
    return;
#endif
    const uint32_t t0 = __float_as_uint(x0) + 0x8000u, t1 = __float_as_uint(x1) + 0x8000u;
    p0 = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(t0 & 0xffff0000u), r1 = x1 - __uint_as_float(t1 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0) + 0x8000u, v1 = __float_as_uint(r1) + 0x8000u;
    p1 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float l0 = r0 - __uint_as_float(v0 & 0xffff0000u), l1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);      // exact: <= 8 significant bits left
}
// The same split of (s x0, s x1) for a sign s = +-1 that is a run-time value of the thread, at the same 15 instructions: adding 2^31
// to the bit pattern flips the sign bit, so the rounding constant k = 0x8000 (s = +1) or 0x80008000 (s = -1) yields the rounded
// first plane of s x directly, and the first residual s x - plane0 is one fma (exact, as above).  The tiled engines stage every
// other A row negated and undo it in the epilogue (x6_dither): the bf16 matrix instruction's small bias toward
// -inf then alternates from row to row instead of being common to all samples.
__device__ __forceinline__ void split2_bf16x3_sg(float x0, float x1, uint32_t k, float s, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    const uint32_t t0 = __float_as_uint(x0) + k, t1 = __float_as_uint(x1) + k;
    p0 = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
    const float r0 = __builtin_fmaf(s, x0, -__uint_as_float(t0 & 0xffff0000u)), r1 = __builtin_fmaf(s, x1, -__uint_as_float(t1 & 0xffff0000u));
    const uint32_t v0 = __float_as_uint(r0) + 0x8000u, v1 = __float_as_uint(r1) + 0x8000u;
    p1 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float l0 = r0 - __uint_as_float(v0 & 0xffff0000u), l1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}
// mrl_set_option "x6_dither" [MRL_X6_DITHER, 3], two bits.  1: forward / data-gradient engines alternate the sign of groups of 8
// staged rows, the weight-gradient engines the sign of every other partial slab (one operand staged negated, the result written
// with the sign undone).  2: the tiled engines permute which row of a group of 8 a thread stages so that their LDS stores are free
// of bank conflicts (gemmx6.hip.h; bit-identical results).
inline int& x6_dither() { static int p = getenv("MRL_X6_DITHER") ? atoi(getenv("MRL_X6_DITHER")) : 3; return p; }
// one value -> the bf16 bit patterns of its three planes (the weight-plane kernels)
__device__ __forceinline__ void split1_bf16x3(float v, uint16_t& b0, uint16_t& b1, uint16_t& b2) {
    uint32_t q0, q1, q2;
    split2_bf16x3(v, 0.f, q0, q1, q2);
    b0 = (uint16_t)q0; b1 = (uint16_t)q1; b2 = (uint16_t)q2;
}

inline int wres_kp(int K) { return (K + 63) / 64 * 64 + 4; }
inline size_t wres_lds_bytes(int zc, int ncols, int K) {
    return (size_t)zc * ((ncols + 31) / 32 * 32) * wres_kp(K) * sizeof(float);
}

template <class AL, class BL, class EF, int NT, int PF, int WAVES>
inline hipError_t launch_wres(const AL& al, const BL& bl, const EF& ef, int zc, int K, int N, long total_tiles,
                              int num_cus, hipStream_t stream) {
    const int KP = wres_kp(K);
    const size_t lds = (size_t)zc * NT * 32 * KP * sizeof(float);
    auto kern = wres_kernel<AL, BL, EF, NT, PF, WAVES>;
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    long want = (total_tiles + WAVES - 1) / WAVES;
    int grid = (int)std::min<long>(std::max<long>(want, 1), num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, stream, al, bl, ef, zc, K, N, KP, total_tiles);
    return hipGetLastError();
}

}  // namespace mrl
