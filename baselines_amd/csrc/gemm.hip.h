// fp32 MFMA implicit-GEMM template for gfx950 (MI355X).
//
//   C[M,N] = sum_k A[M,k] * B[k,N]        (fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32,
//                                           bitwise an fmaf chain -> holds the 1e-5 loss-parity bar)
//
// One kernel serves every dense op of the PPO2 learner (reference: TF ops reached from
// baselines/a2c/utils.py:37-63 `conv`/`fc` and their tf.gradients): conv forward (im2col
// addressing done in the A loader), conv data-gradient (gather form), conv/fc weight-gradient
// (split-K over the batch*pixels dimension), fc forward / data-gradient.  Operands are described
// by loader functors so gather (minibatch indices, ppo2.py:162-164), u8->f32 /255
// (models.py:19) and im2col never materialise in HBM; results leave through epilogue functors
// (bias+activation, activation-derivative masking, split-K partial slabs).
//
// Tiling: 256 threads = 4 waves (WM x WN), each wave TM x TN tiles of 32x32; BK = 32.
// LDS images: "KC" operand (k contiguous in memory) -> S[row][36] read with ds_read_b128
// (row stride 36 floats = conflict-free for the 16-lane b128 groups); "MC" operand (row
// contiguous in memory) -> S[k][rows+4] read with ds_read_b32.  Within every K-block of 8 the
// MFMA step s (0..3) of half-wave h consumes k = 8*kb + 4*h + s on BOTH operands, so a KC
// operand needs one b128 per 4 MFMAs.  Next tile's global loads are issued before the MFMA
// block and written to LDS after it (register-staged software pipeline).
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

// ------------------------------------------------------------------------------------------
// Loader functors.  Interface:
//   static constexpr bool KC;            // true: 4 consecutive k for one row; false: 4 consecutive rows for one k
//   float4 load(int row, int k, int z)   // zero-filled outside [0,rows) x [0,kmax)
// ------------------------------------------------------------------------------------------

// Dense row-major matrix P[row*ld + k], k contiguous.
struct RowKC {
    static constexpr bool KC = true;
    const float* p; long ld; int rows; int kmax; int vec;
    __device__ __forceinline__ float4 load(int row, int k, int) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= rows || k >= kmax) return v;
        const float* q = p + (long)row * ld + k;
        if (vec && k + 3 < kmax) return *reinterpret_cast<const float4*>(q);
        v.x = q[0];
        if (k + 1 < kmax) v.y = q[1];
        if (k + 2 < kmax) v.z = q[2];
        if (k + 3 < kmax) v.w = q[3];
        return v;
    }
};

// Dense matrix P[k*ld + row], row contiguous (the GEMM "k" is the slow index).
struct RowMC {
    static constexpr bool KC = false;
    const float* p; long ld; int rows; int kmax; int vec;
    __device__ __forceinline__ float4 load(int row, int k, int) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= rows || k >= kmax) return v;
        const float* q = p + (long)k * ld + row;
        if (vec && row + 3 < rows) return *reinterpret_cast<const float4*>(q);
        v.x = q[0];
        if (row + 1 < rows) v.y = q[1];
        if (row + 2 < rows) v.z = q[2];
        if (row + 3 < rows) v.w = q[3];
        return v;
    }
};

// Minibatch-gathered observation rows (f32), KC: element (b, k) = obs[srow(b)*ld + k].
// idx holds the reference's env-major flat index i = e*T + t (runner.py:69-74); storage is
// time-major [T][N] so srow = (i % T) * N + i / T.  idx == nullptr: srow = b.
struct GatherRowsBase {
    const float* p; long ld; const int64_t* idx; int T; int N; int rows; int kmax; int vec;
    __device__ __forceinline__ long srow(int b) const {
        if (!idx) return b;
        long i = idx[b];
        return (i % T) * (long)N + i / T;
    }
};
struct GatherKC : GatherRowsBase {
    static constexpr bool KC = true;
    __device__ __forceinline__ float4 load(int row, int k, int) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= rows || k >= kmax) return v;
        const float* q = p + srow(row) * ld + k;
        if (vec && k + 3 < kmax) return *reinterpret_cast<const float4*>(q);
        v.x = q[0];
        if (k + 1 < kmax) v.y = q[1];
        if (k + 2 < kmax) v.z = q[2];
        if (k + 3 < kmax) v.w = q[3];
        return v;
    }
};
// Same data seen as the A' operand of a weight-gradient GEMM: GEMM row = feature (contiguous),
// GEMM k = sample b.
struct GatherMC : GatherRowsBase {
    static constexpr bool KC = false;
    __device__ __forceinline__ float4 load(int row, int k, int) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= kmax || k >= rows) return v;   // here: rows = #samples (GEMM k), kmax = #features (GEMM rows)
        const float* q = p + srow(k) * ld + row;
        if (vec && row + 3 < kmax) return *reinterpret_cast<const float4*>(q);
        v.x = q[0];
        if (row + 1 < kmax) v.y = q[1];
        if (row + 2 < kmax) v.z = q[2];
        if (row + 3 < kmax) v.w = q[3];
        return v;
    }
};

// im2col view of an NHWC image batch (VALID padding): pixel m = (b, oy, ox), conv-k = (ky, kx, c);
// 4 consecutive conv-k never straddle a patch row because C % 4 == 0.
// U8: input is uint8 and is scaled by /255 on load (models.py:19), optionally gathered through idx.
template <bool U8>
struct ConvPatch {
    const void* p; int H, W, C, rf, stride, OH, OW; int npix; int kconv;   // npix = B*OH*OW, kconv = rf*rf*C
    const int64_t* idx; int T; int N;
    __device__ __forceinline__ float4 at(int m, int k) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m >= npix || k >= kconv) return v;
        int ohw = OH * OW;
        int b = m / ohw, r = m - b * ohw;
        int oy = r / OW, ox = r - oy * OW;
        int rowk = rf * C;
        int ky = k / rowk, kr = k - ky * rowk;
        long img = b;
        if (idx) { long i = idx[b]; img = (i % T) * (long)N + i / T; }
        long off = ((img * H + (oy * stride + ky)) * W + ox * stride) * C + kr;
        if (U8) {
            uint32_t u = *reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(p) + off);
            v.x = (float)(u & 0xff) / 255.f;
            v.y = (float)((u >> 8) & 0xff) / 255.f;
            v.z = (float)((u >> 16) & 0xff) / 255.f;
            v.w = (float)(u >> 24) / 255.f;
            return v;
        } else {
            return *reinterpret_cast<const float4*>(static_cast<const float*>(p) + off);
        }
    }
};
template <bool U8> struct ConvPatchKC : ConvPatch<U8> {   // forward: GEMM row = pixel, GEMM k = conv-k
    static constexpr bool KC = true;
    __device__ __forceinline__ float4 load(int row, int k, int) const { return this->at(row, k); }
};
template <bool U8> struct ConvPatchMC : ConvPatch<U8> {   // wgrad: GEMM row = conv-k, GEMM k = pixel
    static constexpr bool KC = false;
    __device__ __forceinline__ float4 load(int row, int k, int) const { return this->at(k, row); }
};

// Data-gradient (gather form) of a VALID strided conv.  Input pixels are enumerated per stride
// parity class z = py*stride + px: row -> (b, yy, xx) with iy = yy*stride + py.  GEMM k = (tap, n),
// tap = (a, b2), contributing output pixel (yy - a, xx - b2) and filter tap (py + stride*a, px + stride*b2).
struct DgradGeom {
    int H, W, C, rf, stride, OH, OW, NF, taps;   // taps per dim = ceil(rf/stride)
    int HY, WX;                                   // class grid extents: ceil(H/stride), ceil(W/stride)
    int B;
};
struct DgradA : DgradGeom {   // KC over n
    static constexpr bool KC = true;
    const float* dz;            // [B, OH, OW, NF]
    __device__ __forceinline__ float4 load(int row, int k, int z) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int per = HY * WX;
        if (row >= B * per || k >= taps * taps * NF) return v;
        int b = row / per, r = row - b * per;
        int yy = r / WX, xx = r - yy * WX;
        int tap = k / NF, n = k - tap * NF;
        int a = tap / taps, b2 = tap - a * taps;
        int oy = yy - a, ox = xx - b2;
        if (oy < 0 || oy >= OH || ox < 0 || ox >= OW) return v;
        return *reinterpret_cast<const float4*>(dz + ((long)(b * OH + oy) * OW + ox) * NF + n);
    }
};
struct DgradB : DgradGeom {   // GEMM row = input channel c, KC over n;  W is HWIO [ky][kx][c][n]
    static constexpr bool KC = true;
    const float* w;
    __device__ __forceinline__ float4 load(int row, int k, int z) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= C || k >= taps * taps * NF) return v;
        int py = z / stride, px = z - py * stride;
        int tap = k / NF, n = k - tap * NF;
        int a = tap / taps, b2 = tap - a * taps;
        int ky = py + stride * a, kx = px + stride * b2;
        if (ky >= rf || kx >= rf) return v;
        return *reinterpret_cast<const float4*>(w + ((long)(ky * rf + kx) * C + row) * NF + n);
    }
};

// ------------------------------------------------------------------------------------------
// Epilogue functors:  void store(int m, int n, float acc, int z)
// ------------------------------------------------------------------------------------------
struct EpiBiasAct {       // out[m*ld + n] = act(acc + bias[n])
    float* out; long ld; const float* bias; int act;
    __device__ __forceinline__ void store(int m, int n, float acc, int) const {
        out[(long)m * ld + n] = act_fwd(acc + bias[n], act);
    }
};
struct EpiMaskAct {       // out[m*ld + n] = acc * act'(h[m*ld + n])      (fc data-gradient)
    float* out; long ld; const float* h; int act;
    __device__ __forceinline__ void store(int m, int n, float acc, int) const {
        long o = (long)m * ld + n;
        out[o] = acc * act_bwd_from_out(h[o], act);
    }
};
struct EpiDgradConv : DgradGeom {   // scatter rows of class z back to NHWC, masked by act'(h_prev)
    float* out; const float* h; int act;
    __device__ __forceinline__ void store(int m, int n, float acc, int z) const {
        int per = HY * WX;
        int b = m / per, r = m - b * per;
        int yy = r / WX, xx = r - yy * WX;
        int py = z / stride, px = z - py * stride;
        int iy = yy * stride + py, ix = xx * stride + px;
        if (iy >= H || ix >= W) return;
        long o = ((long)(b * H + iy) * W + ix) * C + n;
        out[o] = acc * act_bwd_from_out(h[o], act);
    }
};
struct EpiPartial {       // split-K slab: part[z][m*N + n] = acc
    float* part; long slab; int N;
    __device__ __forceinline__ void store(int m, int n, float acc, int z) const {
        part[(long)z * slab + (long)m * N + n] = acc;
    }
};

// ------------------------------------------------------------------------------------------
template <class F, int R>
struct Stage {
    // number of float4 a thread stages per tile
    static constexpr int NV = R / 32;
    // KC image: [R][LDK];  MC image: [BK][R+4]
    static constexpr int ELEMS = F::KC ? R * GEMM_LDK : GEMM_BK * (R + 4);

    __device__ static __forceinline__ void gload(const F& f, float4 (&v)[NV], int r0, int k0, int z, int tid) {
        if (F::KC) {
#pragma unroll
            for (int p = 0; p < NV; ++p) v[p] = f.load(r0 + p * 32 + (tid >> 3), k0 + (tid & 7) * 4, z);
        } else {
            constexpr int V4 = R / 4;            // float4 per k-line
            constexpr int LPP = 256 / V4;        // k-lines per pass
#pragma unroll
            for (int p = 0; p < NV; ++p) v[p] = f.load(r0 + (tid % V4) * 4, k0 + p * LPP + tid / V4, z);
        }
    }
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

template <class AF, class BF, class EF, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(AF af, BF bf, EF ef, int M, int N, int K, int ksplit) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    using SA = Stage<AF, BM>;
    using SB = Stage<BF, BN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + SA::ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
    int kbeg = 0, kend = K;
    if (ksplit < K) { kbeg = z * ksplit; kend = min(K, kbeg + ksplit); }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[SA::NV], rb[SB::NV];
    if (kbeg < kend) {
        SA::gload(af, ra, m0, kbeg, z, tid);
        SB::gload(bf, rb, n0, kbeg, z, tid);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
        __syncthreads();                       // previous tile's fragment reads are done
        SA::swrite(As, ra, tid);
        SB::swrite(Bs, rb, tid);
        __syncthreads();
        if (k0 + GEMM_BK < kend) {             // next tile in flight during the MFMA block
            SA::gload(af, ra, m0, k0 + GEMM_BK, z, tid);
            SB::gload(bf, rb, n0, k0 + GEMM_BK, z, tid);
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
    }
    // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                int col = n0 + (wn * TN + b) * 32 + i;
                if (row < M && col < N) ef.store(row, col, acc[a][b][r], z);
            }
}

template <class AF, class BF, class EF, int WM, int WN, int TM, int TN>
inline hipError_t launch_gemm(const AF& af, const BF& bf, const EF& ef, int M, int N, int K,
                              int zdim, int ksplit, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    size_t lds = (size_t)(Stage<AF, BM>::ELEMS + Stage<BF, BN>::ELEMS) * sizeof(float);
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, zdim);
    if (M <= 0 || N <= 0) return hipSuccess;
    hipLaunchKernelGGL((gemm_kernel<AF, BF, EF, WM, WN, TM, TN>), grid, dim3(256), lds, stream,
                       af, bf, ef, M, N, K, ksplit);
    return hipGetLastError();
}

}  // namespace mrl
