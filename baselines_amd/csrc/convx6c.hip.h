// Class-resident conv forward on the split-bf16 matrix pipe (gfx950, round 5): NatureCNN's conv2 / conv3 (common/models.py:21-22 via
// a2c/utils.py:37-56).  Same products as gemm_x6r_kernel / gemm_x6_kernel (both operands split exactly into three bf16
// planes, six plane products per multiply, fp32 accumulation, x6_dither sign alternation), different STRUCTURE:
//
//   tiled engines (gemmx6*.hip.h): a k step = 32 channels of ONE filter tap; every k step loads its own im2col tile
//       (256 rows x 128 B), writes it to LDS and crosses two barriers -- 16 (conv2) / 18 (conv3) times per tile, and every
//       input element is loaded 4 / 9 times (conv2 forward fetched 16.8 GB for 6.7 GB of input, profiles/r04z_pmc_hbm.json).
//   here: the taps of one stride-parity class (py, px) read the SAME input pixels, shifted by whole positions of the class
//       grid: output pixel (oy, ox), tap (ky, kx) = (py + S a, px + S c) reads grid position (oy + a, ox + c).  A tile is G
//       whole images; per pass (class x 32-channel chunk) the workgroup stages the class grid of its G images ONCE as raw
//       fp32 (G * GH * GW rows of 128 B, 43 / 58 KB) and then runs ALL T*T taps of the class out of it -- 4 (conv2) / 9
//       (conv3) k steps = 192 / 432 MFMAs per wave between two barriers, im2col reduced to an immediate row offset of the
//       fragment read.  Every input element is loaded once, 4 / 2 barrier pairs per tile instead of 16 / 18, the operand
//       split happens on the fragment path between the MFMAs (gemmx6r.hip.h), and the next pass's loads are spread over
//       the MFMA phase one per unit.
//   weights: plane fragments in MFMA operand order, [k step][k block][plane][column block][lane][8 bf16] (x6c_split_planes_kernel), so a
//       wave's B fragment is ONE fully coalesced 1 KB load from the L2-resident 196 / 221 KB tensor -- no LDS, no barrier;
//       double-buffered in registers one k block ahead.
// Transposed-accumulator epilogue (planes.hip.h TrBiasRelu: bias + ReLU + bit mask, 16-byte stores).  Rows of a tile =
// G * NPIX output pixels (243 / 245) padded to 256; two workgroups per CU, persistent over the tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemmx6r.hip.h"

namespace mrl {

template <int H, int W, int C, int RF, int S, int NF, int G>
struct X6cGeom {
    static constexpr int OH = (H - RF) / S + 1, OW = (W - RF) / S + 1, NPIX = OH * OW;
    static constexpr int T = RF / S;                               // taps per dimension and class
    static constexpr int GH = OH + T - 1, GW = OW + T - 1;         // class grid positions a tile's taps reach
    static constexpr int CPC = C / 32, NPASS = S * S * CPC, NTAP = T * T, NSTEP = NPASS * NTAP;
    static constexpr int ROWS = G * GH * GW;                       // LDS rows (128 B of one pixel's 32 channels) per pass
    static constexpr int PITCH = 144;                              // bytes: 128 + 16 (odd multiple of 16: conflict-free b128 reads)
    static constexpr int PIECES = ROWS * 8, NP = (PIECES + 255) / 256;
    static constexpr int MROWS = G * NPIX;                         // real output rows per tile (<= 256)
    static constexpr size_t LDS_BYTES = (size_t)ROWS * PITCH;
    static constexpr long K = (long)RF * RF * C;
    static_assert(RF % S == 0 && C % 32 == 0 && NF == 64, "class-major passes over 32-channel chunks, 64 filters");
    static_assert(S * (GH - 1) + (S - 1) < H + S && MROWS <= 256 && NSTEP * 32 == K, "geometry");
    static_assert((NP - 1) * 32 * PITCH < 65536, "LDS store immediates");
};

// fragment-ordered weight planes: out[((ks * 2 + kb) * 3 + pl) * 2 + cb][lane][8] (bf16 bits), ks = k step in class-major order
// (pass = (py, px, kc) outermost, then the taps (a, c) of the class), lane (i, h) = filter cb * 32 + i, k = 16 kb + 8 h .. + 7
template <int H, int W, int C, int RF, int S, int NF, int G>
__global__ __launch_bounds__(256) void x6c_split_planes_kernel(const float* __restrict__ w, uint16_t* __restrict__ out) {
    using Q = X6cGeom<H, W, C, RF, S, NF, G>;
    const int total = Q::NSTEP * 2 * 2 * 64 * 8;                   // (ks, kb, cb, lane, j); the three planes come from one value
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e & 7, lane = (e >> 3) & 63, cb = (e >> 9) & 1, kb = (e >> 10) & 1, ks = e >> 11;
        const int pass = ks / Q::NTAP, tap = ks - pass * Q::NTAP;
        const int cls = pass / Q::CPC, kc = pass - cls * Q::CPC;
        const int py = cls / S, px = cls - py * S, a = tap / Q::T, c = tap - a * Q::T;
        const int ky = py + S * a, kx = px + S * c;
        const int ch = kc * 32 + kb * 16 + 8 * (lane >> 5) + j, n = cb * 32 + (lane & 31);
        uint16_t b0, b1, b2;
        split1_bf16x3(w[((long)(ky * RF + kx) * C + ch) * NF + n], b0, b1, b2);      // HWIO weights (a2c/utils.py:46)
        const long base = ((long)(ks * 2 + kb) * 3 * 2 + cb) * 512 + lane * 8 + j;  // + pl * 2 * 512
        out[base] = b0; out[base + 1024] = b1; out[base + 2048] = b2;
    }
}

// DBG (timing experiments, wrong results except bit 1; builds with -DMRL_X6_EXPERIMENTS reach them through option conv_x6c = 1 + 2 * DBG):
// 1 = staging loads in the last three units of a pass, 2 = B fragments loaded once per pass, 4 = no MFMAs, 8 = 3-instruction fake split, 16 = no epilogue
template <int H, int W, int C, int RF, int S, int NF, int G, class EF, int VPM, int DBG>
__global__ __launch_bounds__(256, 2) void conv_x6c_kernel(const float* __restrict__ x, const uint16_t* __restrict__ Bf, EF ef, int B,
                                                          int ntiles, int dither) {
    using Q = X6cGeom<H, W, C, RF, S, NF, G>;
    extern __shared__ __attribute__((aligned(16))) uint8_t x6c_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // x6_dither bit 0: rows 8..15 / 24..31 of every 32-row block multiplied negated, un-negated in the epilogue (per-lane sign)
    const bool sg_odd = (dither & 1) && (i & 8);
    const uint32_t sg_k = sg_odd ? 0x80008000u : 0x8000u;
    const float sg_s = sg_odd ? -1.f : 1.f;

    // ---- tile-independent addresses
    // staging piece j of this thread: LDS row (tid >> 3) + 32 j, 16-byte chunk tid & 7; source = class-grid position of image b_l
    // stride 1 with the class grid = the image (conv3): consecutive LDS rows are consecutive pixels, the offsets are linear in j (no registers)
    constexpr bool LINEAR = S == 1 && Q::GH == H && Q::GW == W;
    int goff[LINEAR ? 1 : Q::NP];                      // element offset inside the tile's G images, class (0, 0), chunk kc = 0
    if constexpr (LINEAR) {
        goff[0] = (tid >> 3) * C + (tid & 7) * 4;
    } else {
#pragma unroll
        for (int j = 0; j < Q::NP; ++j) {
            const int row = min((tid >> 3) + 32 * j, Q::ROWS - 1);
            const int bl = row / (Q::GH * Q::GW), r = row - bl * (Q::GH * Q::GW), gy = r / Q::GW, gx = r - gy * Q::GW;
            goff[j] = ((bl * H + S * gy) * W + S * gx) * C + (tid & 7) * 4;
        }
    }
    auto piece_off = [&](int j) {                      // j is a compile-time constant at every call
        if constexpr (LINEAR) return (j + 1) * 32 <= Q::ROWS ? goff[0] + j * 32 * C : min(goff[0] + j * 32 * C, (Q::ROWS - 1) * C + (tid & 7) * 4);
        else return goff[j];
    };
    uint8_t* const sw = x6c_lds + (tid >> 3) * Q::PITCH + (tid & 7) * 16;        // + j * 32 * PITCH
    // fragment reads: lane (i, h) of row block a_ = output row wave * 64 + a_ * 32 + i -> its class-grid row at tap (0, 0)
    int abase[2];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_) {
        const int ml = wave * 64 + a_ * 32 + i;
        const int mc = ml < Q::MROWS ? ml : 0;         // padding rows of the tile: any valid row (never stored)
        const int bl = mc / Q::NPIX, r = mc - bl * Q::NPIX, oy = r / Q::OW, ox = r - oy * Q::OW;
        abase[a_] = ((bl * Q::GH + oy) * Q::GW + ox) * Q::PITCH + h * 32;
    }
    const uint16_t* const bl_ptr = Bf + lane * 8;      // + (((ks * 2 + kb) * 3 + pl) * 2 + cb) * 512

    auto split_frag = [&](const x6r_f4& lo, const x6r_f4& hi, bf16x8 (&f)[3]) {
        u32x4v p0, p1, p2;
        uint32_t a, b, c;
        if constexpr ((DBG & 8) != 0) {
            p0 = u32x4v{__float_as_uint(lo.x) ^ sg_k, __float_as_uint(lo.z), __float_as_uint(hi.x), __float_as_uint(hi.z)};
            p1 = u32x4v{__float_as_uint(lo.y), __float_as_uint(lo.w) ^ sg_k, __float_as_uint(hi.y), __float_as_uint(hi.w)};
            p2 = p0 ^ p1;
            f[0] = __builtin_bit_cast(bf16x8, p0); f[1] = __builtin_bit_cast(bf16x8, p1); f[2] = __builtin_bit_cast(bf16x8, p2);
            return;
        }
        split2_bf16x3_sg(lo.x, lo.y, sg_k, sg_s, a, b, c); p0[0] = a; p1[0] = b; p2[0] = c;
        split2_bf16x3_sg(lo.z, lo.w, sg_k, sg_s, a, b, c); p0[1] = a; p1[1] = b; p2[1] = c;
        split2_bf16x3_sg(hi.x, hi.y, sg_k, sg_s, a, b, c); p0[2] = a; p1[2] = b; p2[2] = c;
        split2_bf16x3_sg(hi.z, hi.w, sg_k, sg_s, a, b, c); p0[3] = a; p1[3] = b; p2[3] = c;
        f[0] = __builtin_bit_cast(bf16x8, p0); f[1] = __builtin_bit_cast(bf16x8, p1); f[2] = __builtin_bit_cast(bf16x8, p2);
    };
    auto mma = [&](const bf16x8& a, const bf16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0); };

    x6r_f4 ra[Q::NP];                                  // staged operand pieces of the NEXT pass (live across passes and tiles)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b0 = tile * G;
        const int gv = min(G, B - b0);                 // images of this tile (the last tile may hold fewer)
        const float* const xt = x + (long)b0 * (H * W * C);
        const int tnext = tile + (int)gridDim.x;       // the last pass of a tile prefetches the first pass of the workgroup's next tile
        const bool has_next = tnext < ntiles;
        const float* const xn = x + (long)(has_next ? tnext : tile) * G * (H * W * C);
        const int gvn = has_next ? min(G, B - tnext * G) : gv;
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        // operand loads of (tile tl_, pass p_): class (py, px), chunk kc.  Rows of images past the batch end re-read image 0 of the tile.
#define X6C_LOAD1(xt_, gv_, p_, j_)                                                                                   \
        {                                                                                                              \
            const int cls_ = (p_) / Q::CPC, kc_ = (p_) - cls_ * Q::CPC;                                                \
            const int po_ = ((cls_ / S) * W + (cls_ % S)) * C + kc_ * 32;                                              \
            int o_ = piece_off(j_);                                                                                         \
            if ((gv_) < G && o_ >= (gv_) * (H * W * C)) o_ -= (o_ / (H * W * C)) * (H * W * C);                        \
            ra[j_] = *reinterpret_cast<const x6r_f4*>((xt_) + po_ + o_);                                               \
        }
        if (tile == (int)blockIdx.x) {                 // first tile of this workgroup: nothing has prefetched its first pass
#pragma unroll
            for (int j = 0; j < Q::NP; ++j) X6C_LOAD1(xt, gv, 0, j)
        }
        for (int pass = 0; pass < Q::NPASS; ++pass) {
            // the pass's first B fragments do not depend on the tile in LDS: requested ahead of the staging barriers
            const uint16_t* bq = bl_ptr + (long)pass * Q::NTAP * 2 * 6 * 512;
            bf16x8 fb[2][2][3];                        // [buffer][cb][pl]
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) fb[0][cb][pl] = *reinterpret_cast<const bf16x8*>(bq + (pl * 2 + cb) * 512);
            __syncthreads();                           // the previous pass's fragment reads are done
#pragma unroll
            for (int j = 0; j < Q::NP; ++j)
                if ((tid >> 3) + 32 * j < Q::ROWS) *reinterpret_cast<x6r_f4*>(sw + j * 32 * Q::PITCH) = ra[j];
            __syncthreads();
            const bool lastp = pass + 1 == Q::NPASS;
            const int pn = lastp ? 0 : pass + 1;       // (no next tile: a harmless re-load of this tile's first pass)
            const float* const xp = lastp ? xn : xt;
            const int gvp = lastp ? gvn : gv;
            __builtin_amdgcn_s_setprio(1);
            // units u = (tap, kb, a_): 12 MFMAs each; the raw fragment of unit u+1 is read and split between the MFMAs of unit u,
            // the B fragments of k block (tap, kb) + 1 are requested while (tap, kb) multiplies, one staging load per unit
            bf16x8 fa[2][3];
            // raw fragments are read AHEAD units before their split (a ds_read consumed in the unit it is issued in stalls the wave's MFMA
            // stream at its wait); two units where the registers allow it (conv3's 13 staged pieces leave no room: it would spill)
            constexpr int AHEAD = Q::NP <= 10 ? 2 : 1;
            x6r_f4 rlo[2], rhi[2];
            auto raw_addr = [&](int u) {
                const int tn = u >> 2, kn = (u >> 1) & 1, an = u & 1;
                return x6c_lds + abase[an] + ((tn / Q::T) * Q::GW + (tn % Q::T)) * Q::PITCH + kn * 64;
            };
            {
                const uint8_t* s = raw_addr(0);
                split_frag(*reinterpret_cast<const x6r_f4*>(s), *reinterpret_cast<const x6r_f4*>(s + 16), fa[0]);
                if constexpr (AHEAD == 2) {
                    const uint8_t* s1 = raw_addr(1);
                    rlo[1] = *reinterpret_cast<const x6r_f4*>(s1);
                    rhi[1] = *reinterpret_cast<const x6r_f4*>(s1 + 16);
                }
            }
            constexpr int NU = Q::NTAP * 4;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int tap = u >> 2, kb = (u >> 1) & 1, a_ = u & 1;
                if (a_ == 0 && u + 2 < NU && !(DBG & 2)) {           // next k block's B fragments -> the other register buffer
                    const int g1 = (u >> 1) + 1;       // = tap * 2 + kb, next
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
                            fb[g1 & 1][cb][pl] = *reinterpret_cast<const bf16x8*>(bq + ((g1 * 3 + pl) * 2 + cb) * 512);
                }
                // next pass's operand loads.  The memory counter retires in order: a wait for B fragments also waits for every
                // staging load issued before them, so staging loads spread over the phase (LATE = 0: one per unit) put HBM latency
                // in front of every k block; LATE = 1 issues them in the last three units, behind the pass's last B-fragment request
                if constexpr (DBG & 1) {
                    constexpr int PER = (Q::NP + 2) / 3;
                    if (u >= NU - 3) {
#pragma unroll
                        for (int j = (u - (NU - 3)) * PER; j < (u - (NU - 3) + 1) * PER; ++j)
                            if (j < Q::NP) X6C_LOAD1(xp, gvp, pn, j)
                    }
                } else {
                    if (u < Q::NP) X6C_LOAD1(xp, gvp, pn, u)
                }
                if (u + AHEAD < NU) {
                    const uint8_t* s = raw_addr(u + AHEAD);
                    rlo[(u + AHEAD) & 1] = *reinterpret_cast<const x6r_f4*>(s);
                    rhi[(u + AHEAD) & 1] = *reinterpret_cast<const x6r_f4*>(s + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int g0 = u >> 1;
#pragma unroll
                for (int b = 0; b < 2; ++b) {          // small terms first, as the tiled engines
                    if constexpr ((DBG & 4) != 0) {
                        acc[a_][b][0] += (float)fa[u & 1][2][0] + (float)fb[g0 & 1][b][0][1] + (float)fa[u & 1][1][2] + (float)fa[u & 1][0][3];
                        continue;
                    }
                    acc[a_][b] = mma(fa[u & 1][2], fb[(DBG & 2) ? 0 : (g0 & 1)][b][0], acc[a_][b]);
                    acc[a_][b] = mma(fa[u & 1][1], fb[(DBG & 2) ? 0 : (g0 & 1)][b][1], acc[a_][b]);
                    acc[a_][b] = mma(fa[u & 1][0], fb[(DBG & 2) ? 0 : (g0 & 1)][b][2], acc[a_][b]);
                    acc[a_][b] = mma(fa[u & 1][1], fb[(DBG & 2) ? 0 : (g0 & 1)][b][0], acc[a_][b]);
                    acc[a_][b] = mma(fa[u & 1][0], fb[(DBG & 2) ? 0 : (g0 & 1)][b][1], acc[a_][b]);
                    acc[a_][b] = mma(fa[u & 1][0], fb[(DBG & 2) ? 0 : (g0 & 1)][b][0], acc[a_][b]);
                }
                if (u + 1 < NU) {
                    split_frag(rlo[(u + 1) & 1], rhi[(u + 1) & 1], fa[(u + 1) & 1]);
#pragma unroll
                    for (int q = 0; q < 12; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
#undef X6C_LOAD1
        // ---- epilogue: transposed accumulators, lane (i, h) owns output row i of its block and 16 columns of each 32-column block
        TrAux aux[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ml = wave * 64 + a * 32 + i;
                const bool valid = ml < gv * Q::NPIX;
                aux[a][b] = ef.load_aux(valid ? ((long)b0 * Q::NPIX + ml) * ef.ld + b * 32 : 0L, b * 32, h, valid);
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ml = wave * 64 + a * 32 + i;
                const bool valid = ml < gv * Q::NPIX;
                tr_block_epilogue(ef, acc[a][b], aux[a][b], valid ? ((long)b0 * Q::NPIX + ml) * ef.ld + b * 32 : 0L, h,
                                  valid && (!(DBG & 16) || acc[a][b][0] == 12345.678f), sg_s);
            }
    }
}

// mrl_set_option "conv_x6c" [MRL_CONV_X6C, 1]: conv2 / conv3 forward of NatureCNN on the class-resident kernel; 0 = tiled engines
inline int& conv_x6c() { static int p = getenv("MRL_CONV_X6C") ? atoi(getenv("MRL_CONV_X6C")) : 1; return p; }

template <int H, int W, int C, int RF, int S, int NF, int G>
inline size_t x6c_plane_elems() { return (size_t)3 * X6cGeom<H, W, C, RF, S, NF, G>::K * NF; }

template <int H, int W, int C, int RF, int S, int NF, int G, class EF>
inline hipError_t launch_conv_x6c(const float* x, const float* w, uint16_t* planes, const EF& ef, int B, int num_cus, hipStream_t stream) {
    using Q = X6cGeom<H, W, C, RF, S, NF, G>;
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL((x6c_split_planes_kernel<H, W, C, RF, S, NF, G>), dim3(64), dim3(256), 0, stream, w, planes);
    auto kern = conv_x6c_kernel<H, W, C, RF, S, NF, G, EF, 5, 0>;
    int slot = 0;
#ifdef MRL_X6_EXPERIMENTS
    switch (conv_x6c() >> 1) {
        case 1: kern = conv_x6c_kernel<H, W, C, RF, S, NF, G, EF, 5, 1>; slot = 1; break;
        case 2: kern = conv_x6c_kernel<H, W, C, RF, S, NF, G, EF, 5, 2>; slot = 2; break;
        case 4: kern = conv_x6c_kernel<H, W, C, RF, S, NF, G, EF, 5, 4>; slot = 3; break;
        case 8: kern = conv_x6c_kernel<H, W, C, RF, S, NF, G, EF, 5, 8>; slot = 4; break;
        case 16: kern = conv_x6c_kernel<H, W, C, RF, S, NF, G, EF, 5, 16>; slot = 5; break;
        case 6: kern = conv_x6c_kernel<H, W, C, RF, S, NF, G, EF, 5, 6>; slot = 6; break;
        case 10: kern = conv_x6c_kernel<H, W, C, RF, S, NF, G, EF, 5, 10>; slot = 7; break;
        default: break;
    }
#endif
    { hipError_t e = raise_lds_limit((const void*)kern); if (e != hipSuccess) return e; }      // once per (device, kernel)
    const int ntiles = (B + G - 1) / G;
    const int grid = std::max(1, std::min(ntiles, 2 * num_cus));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Q::LDS_BYTES, stream, x, planes, ef, B, ntiles, x6_dither());
    return hipGetLastError();
}

}  // namespace mrl
