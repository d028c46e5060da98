// DQN replay slice on gfx950: HBM-resident transition ring, float64
// sum/min segment trees with the reference's heap layout and association order, stratified
// proportional sampling + importance weights, priority update, double-Q TD target + Huber.
//
// Reference map (paths relative to baselines/):
//   ring           deepq/replay_buffer.py:24-43
//   trees          common/segment_tree.py:36-86 (reduce order, point update), :105-131 (prefix-sum descent)
//   PER            deepq/replay_buffer.py:100-115 (add / _sample_proportional), :155-165 (weights),
//                  :169-191 (update_priorities, sequential => last duplicate wins)
//   TD + Huber     deepq/build_graph.py:396-413, common/tf_util.py:39-45
//
// These are latency / HBM-bound integer+f64 kernels (no MFMA): a 2^20-leaf tree pair is 33.6 MB and
// stays resident in L2 / Infinity Cache; a batch touches B*21 nodes per tree.  Tree nodes are read and
// written with agent-scope relaxed atomics (L1 bypass) because a refresh of level L must observe the
// level L-1 values written by other lanes of the same launch.
#include <math.h>

#include "common.hip.h"
#include "powcr.hip.h"

using namespace mrl;

namespace {

__device__ __forceinline__ double tload(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tstore(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- ring ------------------------------------------------------------------------------------
// rows (next_idx + j) % maxsize, j < n, 16 B per lane when the row size allows
template <typename V>
__global__ __launch_bounds__(256) void ring_rows_kernel(V* __restrict__ dst, const V* __restrict__ src, long maxsize,
                                                        long next_idx, long n, int rowv) {
    const long total = n * rowv;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += (long)gridDim.x * 256L) {
        long j = q / rowv;
        int c = (int)(q - j * rowv);
        long slot = (next_idx + j) % maxsize;
        dst[slot * rowv + c] = src[q];
    }
}
template <typename V>
__global__ __launch_bounds__(256) void take_rows_kernel(const V* __restrict__ src, const int32_t* __restrict__ idx,
                                                        V* __restrict__ dst, long B, int rowv) {
    const long total = B * rowv;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += (long)gridDim.x * 256L) {
        long b = q / rowv;
        int c = (int)(q - b * rowv);
        dst[q] = src[(long)idx[b] * rowv + c];
    }
}

// the whole minibatch in one launch: obs_t rows, obs_tp1 rows and (the last B items) the three scalar columns
template <typename V>
__global__ __launch_bounds__(256) void gather_all_kernel(const V* __restrict__ o1, const V* __restrict__ o2, const int32_t* __restrict__ act,
                                                         const float* __restrict__ rew, const float* __restrict__ done,
                                                         const int32_t* __restrict__ idx, V* __restrict__ o1_out, V* __restrict__ o2_out,
                                                         int32_t* __restrict__ act_out, float* __restrict__ rew_out,
                                                         float* __restrict__ done_out, long B, int rowv) {
    const long per = B * rowv, total = 2 * per + B;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += (long)gridDim.x * 256L) {
        if (q >= 2 * per) {
            const long b = q - 2 * per;
            const long i = idx[b];
            act_out[b] = act[i]; rew_out[b] = rew[i]; done_out[b] = done[i];
            continue;
        }
        const bool second = q >= per;
        const long qq = second ? q - per : q;
        const long b = qq / rowv;
        const int c = (int)(qq - b * rowv);
        const long s = (long)idx[b] * rowv + c;
        if (second) o2_out[qq] = o2[s];
        else o1_out[qq] = o1[s];
    }
}

static int pick_unit(int row_bytes, const void* a, const void* b) {
    auto al = [](const void* p, int k) { return ((uintptr_t)p % k) == 0; };
    if (row_bytes % 16 == 0 && al(a, 16) && al(b, 16)) return 16;
    if (row_bytes % 4 == 0 && al(a, 4) && al(b, 4)) return 4;
    return 1;
}
static int ring_rows(void* dst, const void* src, long maxsize, long next_idx, long n, int row_bytes, hipStream_t st) {
    if (n <= 0) return 0;
    int unit = pick_unit(row_bytes, dst, src), rowv = row_bytes / unit;
    int blocks = (int)std::min<long>((n * rowv + 255) / 256, 8192);
    if (unit == 16) hipLaunchKernelGGL(ring_rows_kernel<uint4>, dim3(blocks), dim3(256), 0, st, (uint4*)dst, (const uint4*)src, maxsize, next_idx, n, rowv);
    else if (unit == 4) hipLaunchKernelGGL(ring_rows_kernel<uint32_t>, dim3(blocks), dim3(256), 0, st, (uint32_t*)dst, (const uint32_t*)src, maxsize, next_idx, n, rowv);
    else hipLaunchKernelGGL(ring_rows_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (uint8_t*)dst, (const uint8_t*)src, maxsize, next_idx, n, rowv);
    MRL_LAUNCH_CHECK();
    return 0;
}
static int take_rows(const void* src, const int32_t* idx, void* dst, long B, int row_bytes, hipStream_t st) {
    if (B <= 0) return 0;
    int unit = pick_unit(row_bytes, dst, src), rowv = row_bytes / unit;
    int blocks = (int)std::min<long>((B * rowv + 255) / 256, 8192);
    if (unit == 16) hipLaunchKernelGGL(take_rows_kernel<uint4>, dim3(blocks), dim3(256), 0, st, (const uint4*)src, idx, (uint4*)dst, B, rowv);
    else if (unit == 4) hipLaunchKernelGGL(take_rows_kernel<uint32_t>, dim3(blocks), dim3(256), 0, st, (const uint32_t*)src, idx, (uint32_t*)dst, B, rowv);
    else hipLaunchKernelGGL(take_rows_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (const uint8_t*)src, idx, (uint8_t*)dst, B, rowv);
    MRL_LAUNCH_CHECK();
    return 0;
}

// ---- trees -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tree_fill_kernel(double* __restrict__ sum_tree, double* __restrict__ min_tree, long nodes) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nodes; i += (long)gridDim.x * 256L) {
        sum_tree[i] = 0.0;                                  // segment_tree.py:33 neutral elements
        min_tree[i] = INFINITY;
    }
}

// Batched point update with the reference's SEQUENTIAL semantics (segment_tree.py:76-86 called in a
// loop): leaf j is written unless a later entry names the same leaf; then every touched ancestor is
// recomputed level by level from its two children.  The final tree equals the sequential result
// because an inner node is always exactly op(left, right) of its current children.
// leaf_mode 0: leaf[j] given (f64); 1: leaf = pow(|td[j]| + eps, alpha) computed here (device pow);
// ring_mode: idx == nullptr -> leaves (start + j) % maxsize all get leaf_const / max_priority ** alpha.
//
// Launch shapes (tree_update below):
//   * whole tree, one workgroup (sub_shift < 0): small batches, ring ranges, small trees.  Levels are separated by
//     __syncthreads.
//   * scattered batch of >= 512 entries into a large tree: the bottom `sub_shift` levels are 2^k independent SUBTREES --
//     phase 1 gives each subtree to its own workgroup (it filters the batch for its leaves; no inter-workgroup
//     dependency, 256 CUs issue the scattered 8-byte accesses instead of one), phase 2 (a second launch, one workgroup)
//     recomputes the few levels above the subtree roots densely and folds the running maximum.  One CU's L2 request rate
//     was what the single-workgroup form waited for: 16 k scattered loads per level (batch 4096: 230 -> ~40 us).
// Within a level no two threads write the same node: dense levels name every node once (node = width + t; an untouched
// node is op(left, right) of unchanged children and keeps its value), ring updates walk index ranges, scattered entries
// in deep levels are almost always distinct (entries that do share an ancestor recompute it to the same value).
__global__ __launch_bounds__(1024) void tree_update_kernel(double* sum_tree, double* min_tree, long capacity,
                                                           const int32_t* __restrict__ idx, const double* __restrict__ leaf,
                                                           const float* __restrict__ td, double eps, double alpha,
                                                           double* __restrict__ max_priority, long ring_start,
                                                           long ring_maxsize, double leaf_const, int n, int sub_shift,
                                                           int phase) {
    // 64 KB of LDS: first the duplicate table (open-addressing hash leaf index -> largest entry j that names it), later
    // reused for the running-max reduction
    constexpr int HS = 8192;
    __shared__ __attribute__((aligned(16))) int hs[2 * HS];
    int* hkey = hs;
    int* hval = hs + HS;
    double* smax = reinterpret_cast<double*>(hs);
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool sub = phase == 1;                             // this workgroup owns the leaves i with (i >> sub_shift) == blockIdx.x
    auto mine = [&](long i) { return !sub || (i >> sub_shift) == (long)blockIdx.x; };
    constexpr int TU = 4;
    const long DENSE = (long)TU * nt;
    auto combine = [&](long node) {
        const double a = tload(sum_tree + 2 * node), b = tload(sum_tree + 2 * node + 1);
        const double c = tload(min_tree + 2 * node), d = tload(min_tree + 2 * node + 1);
        tstore(sum_tree + node, __dadd_rn(a, b));
        tstore(min_tree + node, d < c ? d : c);              // python min(a, b): b only if b < a
    };
    if (phase == 2) {
        // ---- levels above the subtree roots, dense; running maximum over ALL entries (replay_buffer.py:191)
        if (td && max_priority) {
            double lmax = 0.0;
            for (int j = tid; j < n; j += nt) lmax = fmax(lmax, fabs((double)td[j]) + eps);
            smax[tid] = lmax;
            __syncthreads();
            for (int s2 = nt >> 1; s2 > 0; s2 >>= 1) {
                if (tid < s2) smax[tid] = fmax(smax[tid], smax[tid + s2]);
                __syncthreads();
            }
            if (tid == 0) max_priority[0] = fmax(max_priority[0], smax[0]);
        }
        for (long width = capacity >> (sub_shift + 1); width >= 1; width >>= 1) {
            for (long node = width + tid; node < 2 * width; node += nt) combine(node);
            __syncthreads();
        }
        return;
    }
    const bool hashed = idx && n <= HS / 2;                 // larger batches: O(n^2) scan below
    if (hashed) {
        for (int e = tid; e < HS; e += nt) { hkey[e] = -1; hval[e] = -1; }
        __syncthreads();
        for (int j = tid; j < n; j += nt) {
            const int key = idx[j];
            if (!mine(key)) continue;
            int slot = (int)(((unsigned)key * 2654435761u) >> 19) & (HS - 1);
            while (true) {
                const int prev = atomicCAS(&hkey[slot], -1, key);
                if (prev == -1 || prev == key) { atomicMax(&hval[slot], j); break; }
                slot = (slot + 1) & (HS - 1);
            }
        }
        __syncthreads();
    }
    double lmax = 0.0;
    for (int j = tid; j < n; j += nt) {
        long i;
        double v;
        bool write = true;
        if (idx) {
            i = idx[j];
            if (!mine(i)) continue;
            if (td) {
                double p = fabs((double)td[j]) + eps;       // deepq.py:302  new_priorities = |td| + eps
                v = mrl::pow_cr(p, alpha);                   // correctly rounded (powcr.hip.h): = Python's ** wherever glibc rounds correctly
                lmax = fmax(lmax, p);
            } else {
                v = leaf[j];
            }
            if (hashed) {                                    // a later duplicate wins
                int slot = (int)(((unsigned)idx[j] * 2654435761u) >> 19) & (HS - 1);
                while (hkey[slot] != (int)i) slot = (slot + 1) & (HS - 1);
                write = hval[slot] == j;
            } else {
                for (int k = j + 1; k < n; ++k)
                    if (idx[k] == i) { write = false; break; }
            }
        } else {
            i = (ring_start + j) % ring_maxsize;
            // device fast path: the leaf is max_priority ** alpha of the RUNNING maximum kept on the device (no host read-back)
            v = max_priority ? mrl::pow_cr(max_priority[0], alpha) : leaf_const;
            // a wrapped ring range cannot name a slot twice unless n > maxsize (rejected on the host)
        }
        if (write) {
            tstore(sum_tree + capacity + i, v);
            tstore(min_tree + capacity + i, v);
        }
    }
    if (td && max_priority && !sub) {                        // replay_buffer.py:191 running max (phase 2 does it for subtrees)
        __syncthreads();                                     // the duplicate table is dead: its LDS becomes smax
        smax[tid] = lmax;
        __syncthreads();
        for (int s2 = nt >> 1; s2 > 0; s2 >>= 1) {
            if (tid < s2) smax[tid] = fmax(smax[tid], smax[tid + s2]);
            __syncthreads();
        }
        if (tid == 0) max_priority[0] = fmax(max_priority[0], smax[0]);
    }
    __syncthreads();
    // ring mode: leaves [ring_start, ring_start + n) mod ring_maxsize = up to two leaf ranges
    const long r0_lo = ring_start, r0_hi = min(ring_start + n, ring_maxsize) - 1;       // inclusive
    const long r1_hi = ring_start + n - ring_maxsize - 1;                                // second range [0, r1_hi] if >= 0
    const long top_width = sub ? (capacity >> sub_shift) : 1;                            // last level this launch computes
    for (long width = capacity >> 1, shift = 1; width >= top_width; width >>= 1, ++shift) {
        if (!sub && width <= DENSE) {
            for (long node = width + tid; node < 2 * width; node += nt) combine(node);
        } else if (!idx) {
            const long lo0 = (capacity + r0_lo) >> shift, hi0 = (capacity + r0_hi) >> shift;
            for (long node = lo0 + tid; node <= hi0; node += nt) combine(node);
            if (r1_hi >= 0) {
                const long lo1 = capacity >> shift, hi1 = (capacity + r1_hi) >> shift;
                for (long node = lo1 + tid; node <= hi1; node += nt)
                    if (node < lo0 || node > hi0) combine(node);
            }
        } else {
            for (int j0 = tid; j0 < n; j0 += TU * nt) {
                long node[TU];
                double a[TU], b[TU], c[TU], d[TU];
                bool on[TU];
#pragma unroll
                for (int u = 0; u < TU; ++u) {
                    const int j = min(j0 + u * nt, n - 1);            // clamped duplicates recompute the last entry's nodes
                    const long i = idx[j];
                    on[u] = mine(i);
                    node[u] = (capacity + i) >> shift;
                }
#pragma unroll
                for (int u = 0; u < TU; ++u)
                    if (on[u]) {
                        a[u] = tload(sum_tree + 2 * node[u]); b[u] = tload(sum_tree + 2 * node[u] + 1);
                        c[u] = tload(min_tree + 2 * node[u]); d[u] = tload(min_tree + 2 * node[u] + 1);
                    }
#pragma unroll
                for (int u = 0; u < TU; ++u)
                    if (on[u]) {
                        tstore(sum_tree + node[u], __dadd_rn(a[u], b[u]));
                        tstore(min_tree + node[u], d[u] < c[u] ? d[u] : c[u]);
                    }
            }
        }
        __syncthreads();
    }
}

// reduce(0, end_excl) of the sum tree with the reference's association order (segment_tree.py:36-49):
// a PREFIX query descends from the root; whenever it continues into a right child the left sibling's
// value is combined as op(left, rest) -- i.e. the result is left_k + (left_{k+1} + (... + deepest)).
__device__ double sum_prefix(const double* tree, long capacity, long end_incl) {
    // path: record the left siblings met on the way down, then fold from the deepest node upward
    long node = 1, lo = 0, hi = capacity - 1;
    (void)lo;
    double sib[64];
    int ns = 0;
    while (end_incl != hi) {                                 // stop when the (prefix) query covers the node exactly
        long mid = (lo + hi) >> 1;
        if (end_incl <= mid) {
            node = 2 * node; hi = mid;
        } else {                                             // split: full left child + prefix of the right child
            sib[ns++] = tload(tree + 2 * node);
            node = 2 * node + 1; lo = mid + 1;
        }
    }
    double acc = tload(tree + node);
    for (int k = ns - 1; k >= 0; --k) acc = __dadd_rn(sib[k], acc);
    return acc;
}

// stratified proportional sampling + importance weights (replay_buffer.py:107-115, 155-165)
__global__ __launch_bounds__(256) void per_sample_kernel(const double* sum_tree, const double* min_tree, long capacity,
                                                         long length, int B, const double* __restrict__ uniforms,
                                                         double beta, int32_t* __restrict__ idx_out,
                                                         double* __restrict__ w_out, float* __restrict__ w32_out) {
    __shared__ double s_total, s_all, s_maxw;
    if (threadIdx.x == 0 && blockIdx.x >= 0) {
        s_total = sum_prefix(sum_tree, capacity, length - 2);       // sum(0, len-1): newest element excluded (quirk)
        s_all = tload(sum_tree + 1);                                 // sum()
        double p_min = tload(min_tree + 1) / s_all;                  // min() / sum()
        s_maxw = mrl::pow_cr(p_min * (double)length, -beta);
    }
    __syncthreads();
    const double every = s_total / (double)B;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256) {
        double mass = __dadd_rn(__dmul_rn(uniforms[i], every), __dmul_rn((double)i, every));
        long node = 1;
        while (node < capacity) {
            double left = tload(sum_tree + 2 * node);
            if (left > mass) node = 2 * node;
            else { mass = __dsub_rn(mass, left); node = 2 * node + 1; }
        }
        long leaf = node - capacity;
        idx_out[i] = (int32_t)leaf;
        double p_sample = tload(sum_tree + node) / s_all;
        double w = mrl::pow_cr(p_sample * (double)length, -beta) / s_maxw;
        if (w_out) w_out[i] = w;
        if (w32_out) w32_out[i] = (float)w;
    }
}

// double-Q TD target, Huber(delta=1), importance-weighted mean and its gradient w.r.t. q_t
__global__ __launch_bounds__(256) void dqn_td_kernel(const float* __restrict__ q_t, const float* __restrict__ q_tp1,
                                                     const float* __restrict__ q_tp1_online, const int32_t* __restrict__ act,
                                                     const float* __restrict__ rew, const float* __restrict__ done,
                                                     const float* __restrict__ w, float gamma, int B, int nA,
                                                     float* __restrict__ td_out, float* __restrict__ dq_out,
                                                     double* __restrict__ part, float* __restrict__ loss_out) {
    __shared__ double sh[4];
    double lsum = 0.0;
    const float invB = 1.f / (float)B;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        const float* qn = q_tp1 + (long)b * nA;
        const float* sel = q_tp1_online ? q_tp1_online + (long)b * nA : qn;
        int best = 0;
        float bv = sel[0];
        for (int j = 1; j < nA; ++j)
            if (sel[j] > bv) { bv = sel[j]; best = j; }      // first max wins (tf.argmax / reduce_max)
        const float q_best = qn[best];
        const float masked = (1.f - done[b]) * q_best;
        const float target = rew[b] + gamma * masked;
        const int a = act[b];
        const float td = q_t[(long)b * nA + a] - target;
        td_out[b] = td;
        const float ad = fabsf(td);
        const float hub = ad < 1.f ? 0.5f * td * td : (ad - 0.5f);
        const float wb = w ? w[b] : 1.f;
        lsum += (double)(wb * hub);
        if (dq_out) {
            const float g = wb * (ad < 1.f ? td : (td > 0.f ? 1.f : -1.f)) * invB;
            for (int j = 0; j < nA; ++j) dq_out[(long)b * nA + j] = (j == a) ? g : 0.f;
        }
    }
    double t = block_sum_256(lsum, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = t;
        if (gridDim.x == 1 && loss_out) loss_out[0] = (float)((0.0 + t) / (double)B);      // one block: no second launch (same arithmetic)
    }
}
__global__ void dqn_td_final_kernel(const double* __restrict__ part, int nblk, int B, float* __restrict__ loss_out) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nblk; ++i) s += part[i];
        loss_out[0] = (float)(s / (double)B);
    }
}

}  // namespace

// ============================================================================================
extern "C" int mrl_replay_insert(void* obs_t_buf, void* obs_tp1_buf, int32_t* act_buf, float* rew_buf, float* done_buf,
                                 long maxsize, long next_idx, int n, int ob_bytes, const void* obs_t,
                                 const void* obs_tp1, const int32_t* act, const float* rew, const float* done,
                                 void* stream) {
    if (!obs_t_buf || !obs_tp1_buf || !act_buf || !rew_buf || !done_buf || !obs_t || !obs_tp1 || !act || !rew || !done)
        return MRL_EINVAL;
    if (maxsize <= 0 || next_idx < 0 || next_idx >= maxsize || n < 0 || n > maxsize || ob_bytes <= 0) return MRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("replay_insert", 0.0, 2.0 * n * (2.0 * ob_bytes + 12.0), st);
    int rc;
    if ((rc = ring_rows(obs_t_buf, obs_t, maxsize, next_idx, n, ob_bytes, st))) return rc;
    if ((rc = ring_rows(obs_tp1_buf, obs_tp1, maxsize, next_idx, n, ob_bytes, st))) return rc;
    if ((rc = ring_rows(act_buf, act, maxsize, next_idx, n, 4, st))) return rc;
    if ((rc = ring_rows(rew_buf, rew, maxsize, next_idx, n, 4, st))) return rc;
    return ring_rows(done_buf, done, maxsize, next_idx, n, 4, st);
}

extern "C" int mrl_replay_gather(const void* obs_t_buf, const void* obs_tp1_buf, const int32_t* act_buf,
                                 const float* rew_buf, const float* done_buf, const int32_t* idx, int B, int ob_bytes,
                                 void* obs_t_out, void* obs_tp1_out, int32_t* act_out, float* rew_out, float* done_out,
                                 void* stream) {
    if (!obs_t_buf || !obs_tp1_buf || !act_buf || !rew_buf || !done_buf || !idx || !obs_t_out || !obs_tp1_out ||
        !act_out || !rew_out || !done_out || B < 0 || ob_bytes <= 0)
        return MRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("replay_gather", 0.0, 2.0 * B * (2.0 * ob_bytes + 12.0) + 4.0 * B, st);
    int rc;
    if (B > 0 && pick_unit(ob_bytes, obs_t_buf, obs_t_out) == 16 && pick_unit(ob_bytes, obs_tp1_buf, obs_tp1_out) == 16) {
        const int rowv = ob_bytes / 16;
        const int blocks = (int)std::min<long>((2L * B * rowv + B + 255) / 256, 16384);
        hipLaunchKernelGGL(gather_all_kernel<uint4>, dim3(blocks), dim3(256), 0, st, (const uint4*)obs_t_buf, (const uint4*)obs_tp1_buf, act_buf,
                           rew_buf, done_buf, idx, (uint4*)obs_t_out, (uint4*)obs_tp1_out, act_out, rew_out, done_out, (long)B, rowv);
        MRL_LAUNCH_CHECK();
        return 0;
    }
    if ((rc = take_rows(obs_t_buf, idx, obs_t_out, B, ob_bytes, st))) return rc;
    if ((rc = take_rows(obs_tp1_buf, idx, obs_tp1_out, B, ob_bytes, st))) return rc;
    if ((rc = take_rows(act_buf, idx, act_out, B, 4, st))) return rc;
    if ((rc = take_rows(rew_buf, idx, rew_out, B, 4, st))) return rc;
    return take_rows(done_buf, idx, done_out, B, 4, st);
}

static bool pow2(long c) { return c > 0 && (c & (c - 1)) == 0; }

extern "C" int mrl_segtree_init(double* sum_tree, double* min_tree, long capacity, void* stream) {
    if (!sum_tree || !min_tree || !pow2(capacity)) return MRL_EINVAL;
    long nodes = 2 * capacity;
    int blocks = (int)std::min<long>((nodes + 255) / 256, 4096);
    hipLaunchKernelGGL(tree_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sum_tree, min_tree, nodes);
    MRL_LAUNCH_CHECK();
    return 0;
}

static int tree_update(double* sum_tree, double* min_tree, long capacity, const int32_t* idx, const double* leaf,
                       const float* td, double eps, double alpha, double* max_priority, long ring_start,
                       long ring_maxsize, double leaf_const, int n, hipStream_t st) {
    if (n <= 0) return 0;
    ProfScope ps("segtree_update", 0.0, 2.0 * n * 16.0 * 21.0, st);
    constexpr int NSUB_LOG2 = 8;                                  // 256 subtrees
    if (idx && n >= 512 && capacity >= (1L << (NSUB_LOG2 + 10))) {
        // scattered batch into a large tree: one workgroup per subtree, then the levels above the subtree roots
        int levels = 0;
        while ((1L << levels) < capacity) ++levels;
        const int sub_shift = levels - NSUB_LOG2;                 // leaves per subtree = 2^sub_shift
        hipLaunchKernelGGL(tree_update_kernel, dim3(1 << NSUB_LOG2), dim3(256), 0, st, sum_tree, min_tree, capacity, idx, leaf, td,
                           eps, alpha, max_priority, ring_start, ring_maxsize, leaf_const, n, sub_shift, 1);
        MRL_LAUNCH_CHECK();
        hipLaunchKernelGGL(tree_update_kernel, dim3(1), dim3(256), 0, st, sum_tree, min_tree, capacity, idx, leaf, td, eps, alpha,
                           max_priority, ring_start, ring_maxsize, leaf_const, n, sub_shift, 2);
        MRL_LAUNCH_CHECK();
        return 0;
    }
    int threads = n >= 1024 ? 1024 : (n > 256 ? 512 : 256);
    hipLaunchKernelGGL(tree_update_kernel, dim3(1), dim3(threads), 0, st, sum_tree, min_tree, capacity, idx, leaf, td, eps,
                       alpha, max_priority, ring_start, ring_maxsize, leaf_const, n, -1, 0);
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrl_segtree_set(double* sum_tree, double* min_tree, long capacity, const int32_t* idx,
                               const double* leaf, int n, void* stream) {
    if (!sum_tree || !min_tree || !pow2(capacity) || !idx || !leaf || n < 0) return MRL_EINVAL;
    return tree_update(sum_tree, min_tree, capacity, idx, leaf, nullptr, 0.0, 0.0, nullptr, 0, 1, 0.0, n, (hipStream_t)stream);
}

extern "C" int mrl_segtree_set_ring(double* sum_tree, double* min_tree, long capacity, long start, long maxsize, int n,
                                    double leaf, void* stream) {
    if (!sum_tree || !min_tree || !pow2(capacity) || maxsize <= 0 || maxsize > capacity || start < 0 || start >= maxsize ||
        n < 0 || n > maxsize)
        return MRL_EINVAL;
    return tree_update(sum_tree, min_tree, capacity, nullptr, nullptr, nullptr, 0.0, 0.0, nullptr, start, maxsize, leaf, n,
                       (hipStream_t)stream);
}

// replay_buffer.py:100-105 `add` on the device fast path: the new slots get max_priority ** alpha with the running maximum
// that mrl_per_update_from_td maintains in device memory
extern "C" int mrl_segtree_set_ring_dev(double* sum_tree, double* min_tree, long capacity, long start, long maxsize, int n,
                                        const double* max_priority, double alpha, void* stream) {
    if (!sum_tree || !min_tree || !pow2(capacity) || maxsize <= 0 || maxsize > capacity || start < 0 || start >= maxsize ||
        n < 0 || n > maxsize || !max_priority)
        return MRL_EINVAL;
    return tree_update(sum_tree, min_tree, capacity, nullptr, nullptr, nullptr, 0.0, alpha, const_cast<double*>(max_priority),
                       start, maxsize, 0.0, n, (hipStream_t)stream);
}

extern "C" int mrl_per_update_from_td(double* sum_tree, double* min_tree, long capacity, const int32_t* idx,
                                      const float* td, double eps, double alpha, double* max_priority, int n,
                                      void* stream) {
    if (!sum_tree || !min_tree || !pow2(capacity) || !idx || !td || !max_priority || n < 0) return MRL_EINVAL;
    return tree_update(sum_tree, min_tree, capacity, idx, nullptr, td, eps, alpha, max_priority, 0, 1, 0.0, n,
                       (hipStream_t)stream);
}

extern "C" int mrl_per_sample(const double* sum_tree, const double* min_tree, long capacity, long length, int B,
                              const double* uniforms, double beta, int32_t* idx_out, double* weights_out,
                              float* weights_f32_out, void* stream) {
    if (!sum_tree || !min_tree || !pow2(capacity) || !uniforms || !idx_out || B <= 0) return MRL_EINVAL;
    if (length < 2 || length > capacity || !(beta > 0.0)) return MRL_EINVAL;   // len 1 recurses forever in the reference
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("per_sample", 0.0, (double)B * (21.0 * 8.0 + 8.0 + 16.0), st);
    // one workgroup computes p_total etc. per block: keep it to a single block up to 1024 samples per
    // 256 threads stride; larger batches use more blocks (each recomputes the three scalars: 60 loads)
    int blocks = std::max(1, std::min((B + 255) / 256, 256));
    hipLaunchKernelGGL(per_sample_kernel, dim3(blocks), dim3(256), 0, st, sum_tree, min_tree, capacity, length, B,
                       uniforms, beta, idx_out, weights_out, weights_f32_out);
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t mrl_dqn_td_scratch_bytes(int B) { return (size_t)std::max(1, std::min((B + 255) / 256, 1024)) * sizeof(double); }

extern "C" int mrl_dqn_td(const float* q_t, const float* q_tp1_target, const float* q_tp1_online, const int32_t* act,
                          const float* rew, const float* done, const float* weights, float gamma, int B, int nA,
                          float* td_out, float* loss_out, float* dq_out, void* scratch, void* stream) {
    if (!q_t || !q_tp1_target || !act || !rew || !done || !td_out || !loss_out || !scratch || B <= 0 || nA <= 0)
        return MRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int blocks = std::max(1, std::min((B + 255) / 256, 1024));
    ProfScope ps("dqn_td", 0.0, (double)B * ((q_tp1_online ? 12.0 : 8.0) * nA + 24.0 + (dq_out ? 4.0 * nA : 0.0)), st);
    hipLaunchKernelGGL(dqn_td_kernel, dim3(blocks), dim3(256), 0, st, q_t, q_tp1_target, q_tp1_online, act, rew, done,
                       weights, gamma, B, nA, td_out, dq_out, (double*)scratch, loss_out);
    MRL_LAUNCH_CHECK();
    if (blocks > 1) {
        hipLaunchKernelGGL(dqn_td_final_kernel, dim3(1), dim3(64), 0, st, (const double*)scratch, blocks, B, loss_out);
        MRL_LAUNCH_CHECK();
    }
    return 0;
}
