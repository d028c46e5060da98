// fp32 MFMA implicit-GEMM template for gfx950 (MI355X).
//
//   C[M,N] = sum_k A[M,k] * B[k,N]        (fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32,
//                                           bitwise an fmaf chain -> holds the 1e-5 loss-parity bar)
//
// One kernel serves every dense op of the PPO2 learner (reference: TF ops reached from
// baselines/a2c/utils.py:37-63 `conv`/`fc` and their tf.gradients): conv forward (im2col
// addressing done in the A loader), conv data-gradient (gather form per stride-parity class),
// conv/fc weight-gradient (split-K over the batch*pixels dimension, bias gradient folded in),
// fc forward / data-gradient.  Operands are described by loader functors so the minibatch gather
// (ppo2.py:162-164), u8->f32 /255 (models.py:19) and im2col never materialise in HBM; results
// leave through epilogue functors (bias+activation, activation-derivative masking, split-K slabs).
//
// Structure (MI355X-first):
//   * 256 threads = 4 waves (WM x WN), one per SIMD; each wave owns TM x TN tiles of 32x32; BK = 32.
//   * every loader keeps PER-THREAD STATE (row bases computed once, the k decomposition advanced
//     incrementally) so the K loop issues ~1 address add per 16-byte load instead of integer
//     divisions: at 64 cycles per fp32 MFMA the VALU budget is ~16 issue slots per MFMA and the
//     im2col index math was what bound the first version of this kernel.
//   * LDS is double buffered (one s_barrier per K tile); the next tile's global loads are issued
//     before the MFMA block and land in LDS after it (register-staged software pipeline).
//   * LDS images: "KC" operand (k contiguous in memory) -> S[row][36], fragments read with
//     ds_read_b128 (row stride 36 floats: conflict-free for the 16-lane b128 groups); "MC" operand
//     (row contiguous in memory) -> S[k][rows+4], fragments read with 4 ds_read_b32.  Within every
//     K-block of 8 the MFMA step s (0..3) of half-wave h consumes k = 8*kb + 4*h + s on BOTH operands.
//   * blockIdx -> tile mapping is XCD-aware: block b runs on XCD b%8, so the N-tiles that share an
//     A row-panel are given consecutive slots on the SAME XCD (shared L2) instead of round-robin.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDK = GEMM_BK + 4;

__device__ __forceinline__ float act_fwd(float x, int act) {
    if (act == ACT_RELU) return x > 0.f ? x : 0.f;
    if (act == ACT_TANH) return tanhf(x);
    return x;
}
// derivative expressed through the layer OUTPUT h (relu: h>0, tanh: 1-h^2)
__device__ __forceinline__ float act_bwd_from_out(float h, int act) {
    if (act == ACT_RELU) return h > 0.f ? 1.f : 0.f;
    if (act == ACT_TANH) return 1.f - h * h;
    return 1.f;
}

// exact unsigned 32-bit division by a runtime constant (Granlund-Montgomery): 4 VALU ops.
struct FastDiv {
    uint32_t m, s1, s2, d;
    static FastDiv make(uint32_t d) {
        FastDiv f;
        f.d = d;
        uint32_t L = 0;
        while ((1ull << L) < d) ++L;
        f.m = (uint32_t)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
        f.s1 = L < 1 ? L : 1;
        f.s2 = L < 1 ? 0 : L - 1;
        return f;
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const {
        uint32_t t = __umulhi(m, n);
        return (t + ((n - t) >> s1)) >> s2;
    }
};

// uint8 pixel -> float(x)/255.f, bit-exact with the IEEE division for all 256 inputs
// (q = x*r; one Newton correction), 3 VALU ops instead of the ~10 of a true divide.
__device__ __forceinline__ float u8_over_255(float x) {
    const float r = 1.f / 255.f;
    float q = __fmul_rn(x, r);
    float e = __fmaf_rn(-q, 255.f, x);
    return __fmaf_rn(e, r, q);
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 load_partial(const float* q, int nvalid) {
    float4 v = f4zero();
    if (nvalid > 0) v.x = q[0];
    if (nvalid > 1) v.y = q[1];
    if (nvalid > 2) v.z = q[2];
    if (nvalid > 3) v.w = q[3];
    return v;
}

// ------------------------------------------------------------------------------------------
// Loader functors.  Interface (NV = float4 staged per thread per K tile = tile_rows / 32):
//   static constexpr bool KC;     true : a thread stages 4 consecutive k of NV rows
//                                        (row = r0 + p*32 + tid/8, k = k0 + (tid%8)*4)
//                                 false: a thread stages 4 consecutive rows of NV k-lines
//                                        (rows r0 + (tid%V4)*4.., k = k0 + p*LPP + tid/V4; V4 = 8*NV, LPP = 32/NV)
//   template<int NV> struct State;
//   init(State&, r0, k0, z, tid)      once per workgroup tile
//   fetch(State&, float4 (&v)[NV])    loads the current K tile (zero-filled outside the operand)
//                                     and advances the state by GEMM_BK
// Rows beyond the operand are clamped / zeroed: they only feed C rows/columns that are never stored.
// ------------------------------------------------------------------------------------------

// Dense row-major operand P[row*ld + k] (k contiguous), optionally row-gathered through srow
// (minibatch gather of f32 observations: element (b, k) = obs[srow[b]*ld + k]).
struct RowKC {
    static constexpr bool KC = true;
    const float* p; long ld; int rows; int kmax; int vec; const int32_t* srow;
    template <int NV> struct State { const float* q[NV]; int k; };
    template <int NV> __device__ __forceinline__ void init(State<NV>& s, int r0, int k0, int, int tid) const {
        s.k = k0 + (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int row = min(r0 + i * 32 + (tid >> 3), rows - 1);
            long sr = srow ? (long)srow[row] : (long)row;
            s.q[i] = p + sr * ld + s.k;
        }
    }
    template <int NV> __device__ __forceinline__ void fetch(State<NV>& s, float4 (&v)[NV]) const {
        if (vec && s.k + 3 < kmax) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(s.q[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = load_partial(s.q[i], kmax - s.k);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) s.q[i] += GEMM_BK;
        s.k += GEMM_BK;
    }
};

// Dense operand P[k*ld + row] (row contiguous; the GEMM k is the slow index), optionally k-gathered
// through srow (weight-gradient view of gathered f32 observations: GEMM k = sample).
struct RowMC {
    static constexpr bool KC = false;
    const float* p; long ld; int rows; int kmax; int vec; const int32_t* srow;
    template <int NV> struct State { const float* q; int k; int nrow; };
    template <int NV> __device__ __forceinline__ void init(State<NV>& s, int r0, int k0, int, int tid) const {
        constexpr int V4 = NV * 8;
        int row = r0 + (tid % V4) * 4;
        s.nrow = max(0, min(4, rows - row));
        s.k = k0 + tid / V4;
        s.q = p + row + (srow ? 0 : (long)s.k * ld);
    }
    template <int NV> __device__ __forceinline__ void fetch(State<NV>& s, float4 (&v)[NV]) const {
        constexpr int LPP = 32 / NV;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int k = s.k + i * LPP;
            v[i] = f4zero();
            if (k < kmax && s.nrow > 0) {
                const float* q = srow ? s.q + (long)srow[k] * ld : s.q + (long)(i * LPP) * ld;
                if (vec && s.nrow == 4) v[i] = *reinterpret_cast<const float4*>(q);
                else v[i] = load_partial(q, s.nrow);
            }
        }
        if (!srow) s.q += (long)GEMM_BK * ld;
        s.k += GEMM_BK;
    }
};

// im2col view of an NHWC image batch (VALID padding): pixel m = (b, oy, ox), conv-k = (ky, kx, c);
// 4 consecutive conv-k never straddle a patch row because C % 4 == 0.
// U8: input is uint8 and is scaled by /255 on load (models.py:19); srow (optional) gathers images.
struct ConvGeom {
    const void* p; int H, W, C, rf, stride, OH, OW; int npix; int kconv;   // npix = B*OH*OW, kconv = rf*rf*C
    int rowk;                                                              // rf*C
    const int32_t* srow;
    // zero padding above / left of the image (SAME convolutions of the DQN `conv_only` net, common/models.py:222-249;
    // taps below / right of the image are out of range too); 0 for the VALID convolutions of NatureCNN.  Only the generic
    // tiled engine (ConvPatchKC / ConvPatchMC) honours it.
    int pad_t = 0, pad_l = 0;
    FastDiv d_ohw, d_ow, d_rowk, d_c;
    void finish() {
        rowk = rf * C;
        d_ohw = FastDiv::make(OH * OW); d_ow = FastDiv::make(OW); d_rowk = FastDiv::make(rowk); d_c = FastDiv::make(C);
    }
    __host__ __device__ bool padded() const { return (pad_t | pad_l) != 0 || (OH - 1) * stride + rf > H || (OW - 1) * stride + rf > W; }
};
// SRC: what a conv loader reads -- 0: fp32 activations; 1: uint8 pixels -> /255; 2: fp32 pixels -> /255
// (`tf.cast(unscaled_images, tf.float32) / 255.` of common/models.py:19 applies to every image dtype)
template <int SRC> __device__ __forceinline__ float4 conv_ld(const void* p, long off) {
    if (SRC == 2) {
        float4 v = *reinterpret_cast<const float4*>(static_cast<const float*>(p) + off);
        v.x = v.x / 255.f; v.y = v.y / 255.f; v.z = v.z / 255.f; v.w = v.w / 255.f;
        return v;
    }
    if (SRC == 1) {
        uint32_t u = *reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p) + off);
        float4 v;
        v.x = u8_over_255((float)(u & 0xff));
        v.y = u8_over_255((float)((u >> 8) & 0xff));
        v.z = u8_over_255((float)((u >> 16) & 0xff));
        v.w = u8_over_255((float)(u >> 24));
        return v;
    } else {
        return *reinterpret_cast<const float4*>(static_cast<const float*>(p) + off);
    }
}
template <int U8> struct ConvPatchKC : ConvGeom {   // forward: GEMM row = pixel, GEMM k = conv-k
    static constexpr bool KC = true;
    template <int NV> struct State { long base[NV]; int iy0[NV], ix0[NV]; int k, ky, kr; };
    template <int NV> __device__ __forceinline__ void init(State<NV>& s, int r0, int k0, int, int tid) const {
        s.k = k0 + (tid & 7) * 4;
        s.ky = (int)d_rowk.div((uint32_t)s.k);
        s.kr = s.k - s.ky * rowk;
        const int ohw = OH * OW;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int m = min(r0 + i * 32 + (tid >> 3), npix - 1);
            int b = (int)d_ohw.div((uint32_t)m), r = m - b * ohw;
            int oy = (int)d_ow.div((uint32_t)r), ox = r - oy * OW;
            long img = srow ? (long)srow[b] : (long)b;
            s.iy0[i] = oy * stride - pad_t;
            s.ix0[i] = ox * stride - pad_l;
            s.base[i] = ((img * H + s.iy0[i]) * W + s.ix0[i]) * C;
        }
    }
    template <int NV> __device__ __forceinline__ void fetch(State<NV>& s, float4 (&v)[NV]) const {
        const long koff = (long)s.ky * W * C + s.kr;
        if (s.k < kconv && padded()) {                   // taps outside the image read zeros
            const int kx = (int)d_c.div((uint32_t)s.kr);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool ok = (unsigned)(s.iy0[i] + s.ky) < (unsigned)H && (unsigned)(s.ix0[i] + kx) < (unsigned)W;
                v[i] = ok ? conv_ld<U8>(p, s.base[i] + koff) : f4zero();
            }
        } else if (s.k < kconv) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = conv_ld<U8>(p, s.base[i] + koff);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = f4zero();
        }
        s.k += GEMM_BK;
        s.kr += GEMM_BK;
        while (s.kr >= rowk) { s.kr -= rowk; ++s.ky; }
    }
};
template <int U8> struct ConvPatchMC : ConvGeom {   // wgrad: GEMM row = conv-k, GEMM k = pixel
    static constexpr bool KC = false;
    template <int NV> struct State { long koff; int m; int b[NV]; int r[NV]; bool rvalid; int ky, kx; };
    template <int NV> __device__ __forceinline__ void init(State<NV>& s, int r0, int k0, int, int tid) const {
        constexpr int V4 = NV * 8, LPP = 32 / NV;
        int row = r0 + (tid % V4) * 4;
        s.rvalid = row < kconv;                       // kconv % 4 == 0: all four or none
        int rc = min(row, kconv - 4);
        int ky = (int)d_rowk.div((uint32_t)rc), kr = rc - ky * rowk;
        s.koff = (long)ky * W * C + kr;
        s.ky = ky;
        s.kx = (int)d_c.div((uint32_t)kr);
        s.m = k0 + tid / V4;
        const int ohw = OH * OW;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int m = s.m + i * LPP;
            s.b[i] = (int)d_ohw.div((uint32_t)m);
            s.r[i] = m - s.b[i] * ohw;
        }
    }
    template <int NV> __device__ __forceinline__ void fetch(State<NV>& s, float4 (&v)[NV]) const {
        constexpr int LPP = 32 / NV;
        const int ohw = OH * OW;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = f4zero();
            if (s.rvalid && s.m + i * LPP < npix) {
                int oy = (int)d_ow.div((uint32_t)s.r[i]), ox = s.r[i] - oy * OW;
                long img = srow ? (long)srow[s.b[i]] : (long)s.b[i];
                const int iy0 = oy * stride - pad_t, ix0 = ox * stride - pad_l;
                if ((unsigned)(iy0 + s.ky) < (unsigned)H && (unsigned)(ix0 + s.kx) < (unsigned)W)
                    v[i] = conv_ld<U8>(p, ((img * H + iy0) * W + ix0) * C + s.koff);
            }
            s.r[i] += GEMM_BK;
            while (s.r[i] >= ohw) { s.r[i] -= ohw; ++s.b[i]; }
        }
        s.m += GEMM_BK;
    }
};

// Data-gradient (gather form) of a VALID strided conv.  Input pixels are enumerated per stride
// parity class z = py*stride + px: row -> (b, yy, xx) with iy = yy*stride + py.  GEMM k = (tap, n),
// tap = (a, b2), contributing output pixel (yy - a, xx - b2) and filter tap (py + stride*a, px + stride*b2).
struct DgradGeom {
    // H, W: extent of the (zero-padded) input the class grid enumerates; Hr, Wr: the real image, which starts at
    // (pad_t, pad_l) of the padded one (SAME convolutions; pad = 0 and Hr = H for VALID)
    int Hr = 0, Wr = 0, pad_t = 0, pad_l = 0;
    int H, W, C, rf, stride, OH, OW, NF, taps;   // taps per dim = ceil(rf/stride)
    int HY, WX;                                   // class grid extents: ceil(H/stride), ceil(W/stride)
    int B;
    FastDiv d_per, d_wx, d_nf, d_taps;
    void finish() {
        d_per = FastDiv::make(HY * WX); d_wx = FastDiv::make(WX); d_nf = FastDiv::make(NF); d_taps = FastDiv::make(taps);
    }
};
struct DgradKState { int k, n, a, b2; };
__device__ __forceinline__ void dgrad_k_init(const DgradGeom& g, DgradKState& s, int k) {
    s.k = k;
    int tap = (int)g.d_nf.div((uint32_t)k);
    s.n = k - tap * g.NF;
    s.a = (int)g.d_taps.div((uint32_t)tap);
    s.b2 = tap - s.a * g.taps;
}
__device__ __forceinline__ void dgrad_k_advance(const DgradGeom& g, DgradKState& s) {
    s.k += GEMM_BK;
    s.n += GEMM_BK;
    while (s.n >= g.NF) {
        s.n -= g.NF;
        if (++s.b2 == g.taps) { s.b2 = 0; ++s.a; }
    }
}
struct DgradA : DgradGeom {   // KC over n
    static constexpr bool KC = true;
    const float* dz;            // [B, OH, OW, NF]
    template <int NV> struct State { DgradKState ks; int pb[NV], yy[NV], xx[NV]; };
    template <int NV> __device__ __forceinline__ void init(State<NV>& s, int r0, int k0, int, int tid) const {
        dgrad_k_init(*this, s.ks, k0 + (tid & 7) * 4);
        const int per = HY * WX;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int m = r0 + i * 32 + (tid >> 3);
            if (m < B * per) {
                int b = (int)d_per.div((uint32_t)m), r = m - b * per;
                s.yy[i] = (int)d_wx.div((uint32_t)r);
                s.xx[i] = r - s.yy[i] * WX;
                s.pb[i] = b * OH * OW;
            } else {
                s.yy[i] = -0x10000; s.xx[i] = 0; s.pb[i] = 0;   // never in range
            }
        }
    }
    template <int NV> __device__ __forceinline__ void fetch(State<NV>& s, float4 (&v)[NV]) const {
        const bool kv = s.ks.k < taps * taps * NF;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int oy = s.yy[i] - s.ks.a, ox = s.xx[i] - s.ks.b2;
            v[i] = f4zero();
            if (kv && (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW)
                v[i] = *reinterpret_cast<const float4*>(dz + (long)(s.pb[i] + oy * OW + ox) * NF + s.ks.n);
        }
        dgrad_k_advance(*this, s.ks);
    }
};
struct DgradB : DgradGeom {   // GEMM row = input channel c, KC over n;  W is HWIO [ky][kx][c][n]
    static constexpr bool KC = true;
    const float* w;
    template <int NV> struct State { DgradKState ks; int c[NV]; int py, px; };
    template <int NV> __device__ __forceinline__ void init(State<NV>& s, int r0, int k0, int z, int tid) const {
        dgrad_k_init(*this, s.ks, k0 + (tid & 7) * 4);
        s.py = z / stride;
        s.px = z - s.py * stride;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = r0 + i * 32 + (tid >> 3);
            s.c[i] = c < C ? c : -1;
        }
    }
    template <int NV> __device__ __forceinline__ void fetch(State<NV>& s, float4 (&v)[NV]) const {
        const int ky = s.py + stride * s.ks.a, kx = s.px + stride * s.ks.b2;
        const bool kv = s.ks.k < taps * taps * NF && ky < rf && kx < rf;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = f4zero();
            if (kv && s.c[i] >= 0)
                v[i] = *reinterpret_cast<const float4*>(w + ((long)(ky * rf + kx) * C + s.c[i]) * NF + s.ks.n);
        }
        dgrad_k_advance(*this, s.ks);
    }
};

// ------------------------------------------------------------------------------------------
// Epilogue functors, three phases so that the kernel can issue ALL auxiliary loads of a tile before the
// first store (a load -> wait -> store chain per element serialises 16 memory round trips per tile):
//   long  addr(m, n, z)        flat output offset, or -1 when the element has no destination
//   float aux(o, n)            the value the store needs from memory (bias / act' source); o >= 0 always
//   void  put(o, acc, aux)     the store
// ------------------------------------------------------------------------------------------
struct EpiBiasAct {       // out[m*ld + n] = act(acc + bias[n])
    static constexpr bool HAS_BIAS = false;
    float* out; long ld; const float* bias; int act;
    uint32_t* mask = nullptr;     // optional ReLU bit mask of the output (gemm_x6_kernel; see WresEpiBiasAct)
    __device__ __forceinline__ bool mask_bit(float acc, float b) const { return act_fwd(acc + b, act) > 0.f; }
    __device__ __forceinline__ long addr(int m, int n, int) const { return (long)m * ld + n; }
    __device__ __forceinline__ float aux(long, int n) const { return bias[n]; }
    __device__ __forceinline__ void put(long o, float acc, float b) const { out[o] = act_fwd(acc + b, act); }
};
struct EpiMaskAct {       // out[m*ld + n] = acc * act'(h[m*ld + n])      (fc data-gradient)
    static constexpr bool HAS_BIAS = false;
    float* out; long ld; const float* h; int act;
    static constexpr uint32_t* mask = nullptr;
    __device__ __forceinline__ bool mask_bit(float, float) const { return false; }
    __device__ __forceinline__ long addr(int m, int n, int) const { return (long)m * ld + n; }
    // h == nullptr: no activation below (store the plain product)
    __device__ __forceinline__ float aux(long o, int) const { return h ? h[o] : 1.f; }
    __device__ __forceinline__ void put(long o, float acc, float hv) const { out[o] = h ? acc * act_bwd_from_out(hv, act) : acc; }
};
struct EpiDgradConv : DgradGeom {   // scatter rows of class z back to NHWC, masked by act'(h_prev)
    static constexpr bool HAS_BIAS = false;
    float* out; const float* h; int act;
    __device__ __forceinline__ long addr(int m, int n, int z) const {
        const int per = HY * WX;
        int b = (int)d_per.div((uint32_t)m), r = m - b * per;
        int yy = (int)d_wx.div((uint32_t)r), xx = r - yy * WX;
        int py = z / stride, px = z - py * stride;
        int iy = yy * stride + py - pad_t, ix = xx * stride + px - pad_l;
        const int hr = Hr ? Hr : H, wr = Wr ? Wr : W;
        if ((unsigned)iy >= (unsigned)hr || (unsigned)ix >= (unsigned)wr) return -1;
        return ((long)(b * hr + iy) * wr + ix) * C + n;
    }
    __device__ __forceinline__ float aux(long o, int) const { return h ? h[o] : 1.f; }
    __device__ __forceinline__ void put(long o, float acc, float hv) const { out[o] = h ? acc * act_bwd_from_out(hv, act) : acc; }
};
// split-K slab of a weight-gradient GEMM: part[z][m*N + n] = acc, followed (same slab) by the bias
// gradient part[z][M*N + n] = sum over this split's rows of dz[row][n] (the B operand's column sums).
struct EpiPartial {
    static constexpr bool HAS_BIAS = true;
    float* part; long slab; int N; long bias_off;
    __device__ __forceinline__ long addr(int m, int n, int z) const { return (long)z * slab + (long)m * N + n; }
    __device__ __forceinline__ float aux(long, int) const { return 0.f; }
    __device__ __forceinline__ void put(long o, float acc, float) const { part[o] = acc; }
    __device__ __forceinline__ void store_bias(int n, float acc, int z) const {
        part[(long)z * slab + bias_off + n] = acc;
    }
};

// split-K partial products of a FORWARD GEMM (no bias column sums): slab z holds sum over k range z of A[m][k] B[k][n]
struct EpiPartialPlain {
    static constexpr bool HAS_BIAS = false;
    float* part; long slab; int N;
    __device__ __forceinline__ long addr(int m, int n, int z) const { return (long)z * slab + (long)m * N + n; }
    __device__ __forceinline__ float aux(long, int) const { return 0.f; }
    __device__ __forceinline__ void put(long o, float acc, float) const { part[o] = acc; }
};

// ------------------------------------------------------------------------------------------
template <class F, int R>
struct Stage {
    static constexpr int NV = R / 32;                    // float4 a thread stages per tile
    // KC image: [R][LDK];  MC image: [BK][R+4]
    static constexpr int ELEMS = F::KC ? R * GEMM_LDK : GEMM_BK * (R + 4);
    using State = typename F::template State<NV>;

    __device__ static __forceinline__ void swrite(float* s, const float4 (&v)[NV], int tid) {
        if (F::KC) {
#pragma unroll
            for (int p = 0; p < NV; ++p)
                *reinterpret_cast<float4*>(s + (p * 32 + (tid >> 3)) * GEMM_LDK + (tid & 7) * 4) = v[p];
        } else {
            constexpr int V4 = R / 4;
            constexpr int LPP = 256 / V4;
#pragma unroll
            for (int p = 0; p < NV; ++p)
                *reinterpret_cast<float4*>(s + (p * LPP + tid / V4) * (R + 4) + (tid % V4) * 4) = v[p];
        }
    }
    // fragment for rows [rb, rb+32), K-block kb: f[s] = S(row = rb + i, k = 8*kb + 4*h + s)
    __device__ static __forceinline__ float4 frag(const float* s, int rb, int kb, int i, int h) {
        if (F::KC) {
            return *reinterpret_cast<const float4*>(s + (rb + i) * GEMM_LDK + kb * 8 + h * 4);
        } else {
            const float* q = s + (kb * 8 + h * 4) * (R + 4) + rb + i;
            return make_float4(q[0], q[R + 4], q[2 * (R + 4)], q[3 * (R + 4)]);
        }
    }
};

template <class AF, class BF, class EF, int WM, int WN, int TM, int TN, bool DB>
__global__ __launch_bounds__(256) void gemm_kernel(AF af, BF bf, EF ef, int M, int N, int K, int ksplit,
                                                   int mtiles, int ntiles, int zdim) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    using SA = Stage<AF, BM>;
    using SB = Stage<BF, BN>;
    constexpr int BUF = SA::ELEMS + SB::ELEMS;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // XCD-aware tile mapping.  A "panel" p = (z, m tile) owns the N tiles that share its A row-panel;
    // panel p lives on XCD p % 8 and its N tiles take consecutive slots there (shared L2), while
    // consecutive panels (m tiles of one z, then the next z) round-robin over the 8 XCDs.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt_i = slot % ntiles;
    const long panel = (long)(slot / ntiles) * 8 + xcd;
    if (panel >= (long)mtiles * zdim) return;
    const int z = (int)(panel / mtiles), mt_i = (int)(panel - (long)z * mtiles);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = mt_i * BM, n0 = nt_i * BN;
    int kbeg = 0, kend = K;
    if (ksplit < K) { kbeg = z * ksplit; kend = min(K, kbeg + ksplit); }
    const int ntile = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum = 0.f;     // bias-gradient column sum (EF::HAS_BIAS, m tile 0, thread = column)

    typename SA::State sa;
    typename SB::State sb;
    float4 ra[SA::NV], rb[SB::NV];
    if (ntile > 0) {
        af.template init<SA::NV>(sa, m0, kbeg, z, tid);
        bf.template init<SB::NV>(sb, n0, kbeg, z, tid);
        af.template fetch<SA::NV>(sa, ra);
        bf.template fetch<SB::NV>(sb, rb);
        if (DB) {
            SA::swrite(smem, ra, tid);
            SB::swrite(smem + SA::ELEMS, rb, tid);
        }
    }
    if (DB) __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const float* As = smem + (DB ? (t & 1) * BUF : 0);
        const float* Bs = As + SA::ELEMS;
        const bool more = t + 1 < ntile;
        if (!DB) {
            __syncthreads();                   // previous tile's fragment reads are done
            SA::swrite(smem, ra, tid);
            SB::swrite(smem + SA::ELEMS, rb, tid);
            __syncthreads();
        }
        if (more) {                            // next tile in flight during the MFMA block
            af.template fetch<SA::NV>(sa, ra);
            bf.template fetch<SB::NV>(sb, rb);
        }
#pragma unroll
        for (int kb = 0; kb < GEMM_BK / 8; ++kb) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = SA::frag(As, (wm * TM + a) * 32, kb, i, h);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = SB::frag(Bs, (wn * TN + b) * 32, kb, i, h);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
                }
        }
        if (EF::HAS_BIAS && !BF::KC && mt_i == 0) {
            // B image is [k][BN+4]: column sums in k order.  Spread over the 4 waves (8 k each) so no
            // single wave carries a 32-deep dependent chain; combined in fixed order at the end.
            const int c = tid % BN, part = tid / BN;          // BN in {32,64,128}: 256/BN parts
            constexpr int PARTS = 256 / BN, KP = GEMM_BK / PARTS;
            float t0 = 0.f;
#pragma unroll
            for (int kk = 0; kk < KP; ++kk) t0 += Bs[(part * KP + kk) * (BN + 4) + c];
            bsum += t0;
        }
        if (DB) {
            if (more) {
                float* An = smem + ((t + 1) & 1) * BUF;
                SA::swrite(An, ra, tid);
                SB::swrite(An + SA::ELEMS, rb, tid);
            }
            __syncthreads();
        }
    }
    // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + (wn * TN + b) * 32 + i;
            const int colc = min(col, N - 1);
            long o[16];
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                o[r] = (row < M && col < N) ? ef.addr(row, col, z) : -1;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = ef.aux(o[r] < 0 ? 0 : o[r], colc);     // all loads first, unconditional
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (o[r] >= 0) ef.put(o[r], acc[a][b][r], x[r]);
        }
    if constexpr (EF::HAS_BIAS) {
        if (!BF::KC && mt_i == 0) {            // combine the per-part column sums in fixed order
            __syncthreads();
            smem[tid] = bsum;
            __syncthreads();
            if (tid < BN && n0 + tid < N) {
                float t = 0.f;
                for (int q = 0; q < 256 / BN; ++q) t += smem[q * BN + tid];
                ef.store_bias(n0 + tid, t, z);
            }
        }
    }
}

template <class AF, class BF, class EF, int WM, int WN, int TM, int TN, bool DB>
inline hipError_t launch_gemm(const AF& af, const BF& bf, const EF& ef, int M, int N, int K,
                              int zdim, int ksplit, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const size_t lds = (DB ? 2 : 1) * (size_t)(Stage<AF, BM>::ELEMS + Stage<BF, BN>::ELEMS) * sizeof(float);
    if (M <= 0 || N <= 0 || zdim <= 0) return hipSuccess;
    const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
    const long panels = (long)mtiles * zdim;
    const long blocks = (panels + 7) / 8 * 8 * ntiles;
    if (blocks > 0x7fffffffL) return hipErrorInvalidValue;
    auto kern = gemm_kernel<AF, BF, EF, WM, WN, TM, TN, DB>;
    if (lds > 64 * 1024) {
        { hipError_t e = raise_lds_limit((const void*)kern, (int)lds); if (e != hipSuccess) return e; }      // once per (device, kernel)
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, af, bf, ef, M, N, K, ksplit,
                       mtiles, ntiles, zdim);
    return hipGetLastError();
}

}  // namespace mrl
