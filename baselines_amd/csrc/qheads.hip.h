// The dueling heads of the DQN Q-network at learner / actor batch sizes, as four kernels (round 6).
//
// deepq/models.py:5-45: action_value = fc(nact)(relu(fc(h)(latent))), state_value = fc(1)(relu(fc(h)(latent))), q = V + (A - mean_a A).
// Layer by layer on the tile engines that was, per pass, av0 (split-K) + its bias/activation pass + av1 + sv0 + pass + sv1 + the dueling
// combine = 7 launches of 5-16 us on a handful of workgroups (80 of a 145 us forward pass at batch 32 / 64) and 13 more going back.  With
// the standard head shape (one hidden layer per head, no layer norm) the work is:
//   q_heads_fwd_kernel    hidden pre-activations of BOTH heads as one [B, K] x [K, N_a + N_s] product on the fp32 matrix pipe
//                         (v_mfma_f32_32x32x2_f32), split over K across workgroups and waves; partials [S][B][N] in the workspace
//   q_heads_out_kernel    per sample: sum of the partials + bias + ReLU -> hidden activations; the two output layers; the dueling combine
//   q_heads_bwd_kernel    per sample: dueling backward, dz of the hidden layers; two more workgroups: the output layers' weight gradients
//   q_heads_wgrad_kernel  dW0[k][n] = sum_b latent[b][k] dz0[b][n] for both heads, written straight into the flat gradient (no slabs:
//                         nothing to split at 32 rows), + the hidden biases' gradients
// and q_lat_dgrad_kernel (qnet.hip.h) takes both heads' dz0 into the latent.  Every sum has a fixed order (deterministic).
// The MFMA operand trick used throughout: lane (r, half) of v_mfma_f32_32x32x2_f32 supplies A[row r][kk = half] and B[kk = half][col r];
// WHICH reduction index a (step, half) pair stands for is free as long as both operands agree -- so operands whose reduction index is
// contiguous in memory are loaded 16 bytes at a time (half h takes elements 8c + 4h .. 8c + 4h + 3 of chunk c, one per MFMA step).
#pragma once

struct QHeads {
    const float* W0[2]; const float* b0[2]; const float* W1[2]; const float* b1[2];      // [K][N0], [N0], [N0][nout], [nout]
    int N0[2], nout[2];
    int nheads, K;
};

// ---- forward: hidden pre-activation partials ---------------------------------------------------------------------------------------
// grid (n tiles of 32 over N0[0] + N0[1], S splits of K, row groups of 32 MT); 8 waves; part[s][b][n], n over both heads
// Two independent problems in one launch (the DQN step's online and target passes): row groups [0, zg0) of grid.z belong to p0.
struct QHeadsFwdProb { QHeads hd; const float* lat; int B; float* part; };
template <int MT>
__global__ __launch_bounds__(512) void q_heads_fwd_kernel(QHeadsFwdProb p0, QHeadsFwdProb p1, int zg0, int cps /* 8-k chunks per split */) {
    __shared__ float red[8][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hf = lane >> 5;
    // (every field is selected by value: a reference into a by-value kernel argument -- or a run-time index into one of its arrays --
    // makes the compiler spill the struct to scratch and address it with flat loads)
    const bool second = (int)blockIdx.z >= zg0;
    const float* __restrict__ lat = second ? p1.lat : p0.lat;
    float* __restrict__ part = second ? p1.part : p0.part;
    const int B = second ? p1.B : p0.B;
    const int N0a = second ? p1.hd.N0[0] : p0.hd.N0[0], N0s = second ? p1.hd.N0[1] : p0.hd.N0[1];
    const int nheads = second ? p1.hd.nheads : p0.hd.nheads, K = second ? p1.hd.K : p0.hd.K;
    const int ntot = N0a + (nheads > 1 ? N0s : 0);
    int n0 = blockIdx.x * 32;
    const bool head = n0 >= N0a;
    if (head) n0 -= N0a;
    const int N = head ? N0s : N0a;
    const float* __restrict__ Wbase = head ? (second ? p1.hd.W0[1] : p0.hd.W0[1]) : (second ? p1.hd.W0[0] : p0.hd.W0[0]);
    const float* W = Wbase + n0 + r;
    const int m0 = ((int)blockIdx.z - (second ? zg0 : 0)) * 32 * MT;
    const int kc = K >> 3;
    const int sb = blockIdx.y * cps, se = min(kc, sb + cps);
    const int per = (se - sb + 7) / 8, cb = sb + wave * per, ce = min(se, cb + per);
    const float* ar[MT];
    bool alive[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        alive[t] = m0 + 32 * t + r < B;
        ar[t] = lat + (long)(m0 + 32 * t + r) * K + 4 * hf;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // chunks in groups of U, the next group's loads issued before the current group's MFMAs (two register sets, ping-pong)
    constexpr int U = 4;
    float4 fa[2][U][MT];
    float fb[2][U][4];
    auto load = [&](int set, int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = c + u < ce;
            const long k = 8L * (c + u) + 4 * hf;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[set][u][j] = mrl::ldz1(W + (k + j) * N, Wbase, live);          // branch-free (convskinny.hip.h)
#pragma unroll
            for (int t = 0; t < MT; ++t) fa[set][u][t] = mrl::ldz4(ar[t] + 8L * (c + u), lat, live && alive[t]);
        }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u][t].x, fb[set][u][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u][t].y, fb[set][u][1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u][t].z, fb[set][u][2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u][t].w, fb[set][u][3], acc[t], 0, 0, 0);
            }
    };
    load(0, cb);                                   // (every load is predicated on its own chunk: loads past `ce` fetch nothing new)
    for (int c = cb; c < ce; c += 2 * U) {
        load(1, c + U);
        mma(0);
        load(0, c + 2 * U);
        if (c + U < ce) mma(1);                    // wave-uniform; no load inside a branch -> exact wait counts
    }
    const int ncol = blockIdx.x * 32;       // column in the concatenated hidden vector
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        if (t) __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][(i >> 2) * 8 + hf * 4 + (i & 3)][r] = acc[t][i];      // row = sample, column = hidden unit
        __syncthreads();
        for (int idx = threadIdx.x; idx < 1024; idx += 512) {
            const int b = idx >> 5, n = idx & 31;
            if (m0 + 32 * t + b >= B) continue;
            float v = red[0][b][n];
#pragma unroll
            for (int w = 1; w < 8; ++w) v += red[w][b][n];
            part[((long)blockIdx.y * B + m0 + 32 * t + b) * ntot + ncol + n] = v;
        }
    }
}

// deterministic block sums of eight floats at once over 256 threads (wave shuffles, then the four wave sums in order; one barrier pair);
// valid in every thread
__device__ __forceinline__ void q_block_sum8(float (&v)[8], float (*sh)[8] /* [4][8] */) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[j] += __shfl_xor(v[j], off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[threadIdx.x >> 6][j] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ((sh[0][j] + sh[1][j]) + sh[2][j]) + sh[3][j];
}

// ---- forward: per sample, hidden activations + output layers + dueling combine ---------------------------------------------------------
// grid B, 256 threads; h0a / h0s: [B][N0] hidden activations (kept for the backward pass); oa / os: the heads' raw outputs
constexpr int QH_MAXOUT = 32;
struct QHeadsOutProb { QHeads hd; const float* part; int S, B; float *h0a, *h0s, *oa, *os, *q; };
__global__ __launch_bounds__(256) void q_heads_out_kernel(QHeadsOutProb p0, QHeadsOutProb p1) {
    __shared__ float sh[4][8];
    __shared__ float outs[QH_MAXOUT + 1];
    const bool second = (int)blockIdx.x >= p0.B;
#define QSEL(f) (second ? p1.f : p0.f)
    const float* __restrict__ part = QSEL(part);
    const int S = QSEL(S), B = QSEL(B);
    float* __restrict__ h0a = QSEL(h0a);
    float* __restrict__ h0s = QSEL(h0s);
    float* __restrict__ oa = QSEL(oa);
    float* __restrict__ os = QSEL(os);
    float* __restrict__ q = QSEL(q);
    const int b = (int)blockIdx.x - (second ? p0.B : 0);
    const int nheads = QSEL(hd.nheads), N0a = QSEL(hd.N0[0]), N0s = QSEL(hd.N0[1]);
    const int ntot = N0a + (nheads > 1 ? N0s : 0);
    for (int head = 0; head < nheads; ++head) {
        const int N = head ? N0s : N0a, no = head ? QSEL(hd.nout[1]) : QSEL(hd.nout[0]), col0 = head ? N0a : 0;
        const float* __restrict__ b0 = head ? QSEL(hd.b0[1]) : QSEL(hd.b0[0]);
        const float* __restrict__ W1 = head ? QSEL(hd.W1[1]) : QSEL(hd.W1[0]);
        const float* __restrict__ b1 = head ? QSEL(hd.b1[1]) : QSEL(hd.b1[0]);
        float* hout = head ? h0s : h0a;
        for (int j0 = 0; j0 < no; j0 += 8) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = 0.f;
            for (int n = threadIdx.x; n < N; n += 256) {
                const float* pp = part + (long)b * ntot + col0 + n;
                float v = 0.f;
                int s = 0;
                for (; s + 4 <= S; s += 4) {              // four partials in flight, summed in split order
                    const float x0 = pp[(long)s * B * ntot], x1 = pp[(long)(s + 1) * B * ntot], x2 = pp[(long)(s + 2) * B * ntot],
                                x3 = pp[(long)(s + 3) * B * ntot];
                    v = (((v + x0) + x1) + x2) + x3;
                }
                for (; s < S; ++s) v += pp[(long)s * B * ntot];
                v = fmaxf(v + b0[n], 0.f);
                if (j0 == 0) hout[(long)b * N + n] = v;
                const float* w1 = W1 + (long)n * no + j0;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j0 + j < no) p[j] += v * w1[j];
            }
            q_block_sum8(p, sh);
            float mine = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if ((int)threadIdx.x == j) mine = p[j];
            if (threadIdx.x < 8 && j0 + (int)threadIdx.x < no)
                outs[head ? QH_MAXOUT : j0 + threadIdx.x] = mine + b1[j0 + threadIdx.x];
        }
    }
    __syncthreads();
    const int nA = QSEL(hd.nout[0]);
#undef QSEL
    if (threadIdx.x < nA) {
        const float a = outs[threadIdx.x];
        oa[(long)b * nA + threadIdx.x] = a;
        float qv = a;
        if (nheads > 1) {
            float s = 0.f;
            for (int j = 0; j < nA; ++j) s += outs[j];
            const float mean = s / (float)nA;
            qv = outs[QH_MAXOUT] + (a - mean);
            if (threadIdx.x == 0) os[b] = outs[QH_MAXOUT];
        }
        q[(long)b * nA + threadIdx.x] = qv;
    }
}

// ---- backward: dueling + output layers --------------------------------------------------------------------------------------------------
// blocks 0 .. B-1: dz of the output layers (dA = dq - mean_a dq, dV = sum_a dq) and of the hidden layers (masked by ReLU'), one sample each;
// blocks B, B+1: the weight / bias gradients of the two output layers (sum over the samples in order)
__global__ __launch_bounds__(256) void q_heads_bwd_kernel(QHeads hd, const float* __restrict__ dq, int B, const float* __restrict__ h0a,
                                                          const float* __restrict__ h0s, float* __restrict__ dz1a, float* __restrict__ dz1s,
                                                          float* __restrict__ dz0a, float* __restrict__ dz0s, float* __restrict__ gW1a,
                                                          float* __restrict__ gb1a, float* __restrict__ gW1s, float* __restrict__ gb1s) {
    const int nA = hd.nout[0];
    const bool duel = hd.nheads > 1;
    if ((int)blockIdx.x < B) {
        __shared__ float da[QH_MAXOUT + 1];
        const int b = blockIdx.x;
        if (threadIdx.x == 0) {
            const float* d = dq + (long)b * nA;
            float s = 0.f;
            for (int j = 0; j < nA; ++j) s += d[j];
            const float mean = s / (float)nA;
            for (int j = 0; j < nA; ++j) {
                const float v = duel ? d[j] - mean : d[j];
                da[j] = v;
                dz1a[(long)b * nA + j] = v;
            }
            da[QH_MAXOUT] = s;
            if (duel) dz1s[b] = s;
        }
        __syncthreads();
        for (int n = threadIdx.x; n < hd.N0[0]; n += 256) {
            const float* w1 = hd.W1[0] + (long)n * nA;
            float acc = 0.f;
            for (int j = 0; j < nA; ++j) acc += da[j] * w1[j];
            const long o = (long)b * hd.N0[0] + n;
            dz0a[o] = acc * act_bwd_from_out(h0a[o], ACT_RELU);
        }
        if (duel)
            for (int n = threadIdx.x; n < hd.N0[1]; n += 256) {
                const long o = (long)b * hd.N0[1] + n;
                dz0s[o] = (da[QH_MAXOUT] * hd.W1[1][n]) * act_bwd_from_out(h0s[o], ACT_RELU);
            }
        return;
    }
    const int head = blockIdx.x - B;
    extern __shared__ float dsh[];                  // [B][no] dz of this head's output layer, recomputed from dq
    const int no = hd.nout[head], N = hd.N0[head];
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* d = dq + (long)b * nA;
        float s = 0.f;
        for (int j = 0; j < nA; ++j) s += d[j];
        const float mean = s / (float)nA;
        if (head == 0)
            for (int j = 0; j < nA; ++j) dsh[b * nA + j] = duel ? d[j] - mean : d[j];
        else
            dsh[b] = s;
    }
    __syncthreads();
    const float* h0 = head ? h0s : h0a;
    float* gW = head ? gW1s : gW1a;
    float* gb = head ? gb1s : gb1a;
    for (int n = threadIdx.x; n < N; n += 256)
        for (int j0 = 0; j0 < no; j0 += 8) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            int b = 0;
            for (; b + 8 <= B; b += 8) {                 // eight activations in flight, accumulated in sample order
                float hv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) hv[u] = h0[(long)(b + u) * N + n];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j0 + j < no) acc[j] += hv[u] * dsh[(b + u) * no + j0 + j];
            }
            for (; b < B; ++b) {
                const float hv = h0[(long)b * N + n];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j0 + j < no) acc[j] += hv * dsh[b * no + j0 + j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j0 + j < no) gW[(long)n * no + j0 + j] = acc[j];
        }
    if ((int)threadIdx.x < no) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dsh[b * no + threadIdx.x];
        gb[threadIdx.x] = s;
    }
}

// ---- backward: the hidden layers' weight gradients ------------------------------------------------------------------------------------------
// grid (k tiles of 32, groups of 4 n tiles over both heads); wave = one 32 x 32 tile of dW0: rows = latent units (A = latent^T, lane r reads
// latent[b][k0 + r]: contiguous across the half-wave), columns = hidden units (B = dz0), the reduction runs over the samples
__global__ __launch_bounds__(256) void q_heads_wgrad_kernel(QHeads hd, const float* __restrict__ lat, const float* __restrict__ dz0a,
                                                            const float* __restrict__ dz0s, int B, float* __restrict__ gW0a,
                                                            float* __restrict__ gb0a, float* __restrict__ gW0s, float* __restrict__ gb0s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hf = lane >> 5;
    const int k0 = blockIdx.x * 32;
    int n0 = (blockIdx.y * 4 + wave) * 32, head = 0;
    const int ntot = hd.N0[0] + (hd.nheads > 1 ? hd.N0[1] : 0);
    if (n0 >= ntot) return;
    if (n0 >= hd.N0[0]) { head = 1; n0 -= hd.N0[0]; }
    const int N = hd.N0[head];
    const float* dz = (head ? dz0s : dz0a) + n0 + r;
    const bool krow = k0 + r < hd.K;
    const float* lp = lat + k0 + r;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    for (int b0 = 0; b0 < B; b0 += 32) {
        float fa[16], fb[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int b = b0 + 2 * i + hf;
            const bool live = b < B;
            fa[i] = mrl::ldz1(lp + (long)b * hd.K, lat, live && krow);
            fb[i] = mrl::ldz1(dz + (long)b * N, dz0a, live);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[i], acc, 0, 0, 0);
            bsum += fb[i];
        }
    }
    float* gW = (head ? gW0s : gW0a) + n0 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = k0 + (i >> 2) * 8 + hf * 4 + (i & 3);
        if (k < hd.K) gW[(long)k * N] = acc[i];
    }
    if (blockIdx.x == 0) {
        bsum += __shfl_xor(bsum, 32, 64);           // even samples (half 0) + odd samples (half 1)
        if (hf == 0) (head ? gb0s : gb0a)[n0 + r] = bsum;
    }
}
