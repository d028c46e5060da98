// DQN Q-network on gfx950: layout + forward / TD-gradient orchestration + per-variable gradient
// clipping and Adam.  Included at the end of model.hip: it re-uses that file's layer machinery (Layer / Net / NetWs,
// layer_forward, net_backward, the GEMM engines and the deterministic slab reductions).
//
// Reference map (paths relative to baselines/):
//   q_func            deepq/models.py:5-45 `build_q_func(network, hiddens=[256], dueling=True)`:
//                       latent = network(X) -> flatten -> action_value: [fc(h) -> relu]* -> fc(num_actions)
//                       [dueling] state_value: [fc(h) -> relu]* -> fc(1);  q = V + (A - mean_a A)
//   networks          common/models.py:74-103 `mlp`, :15-26 `nature_cnn` ('cnn'), :222-249 `conv_only`
//                       (tf.contrib.layers.convolution2d: SAME padding, ReLU, xavier-uniform weights, zero biases;
//                       heads: tf.contrib.layers.fully_connected, same defaults)
//   TD graph          deepq/build_graph.py:380-421: q_t online on obs_t; q_tp1 TARGET net on obs_tp1; double-Q argmax
//                       from the ONLINE net on obs_tp1; Huber(td) weighted mean (mrl_dqn_td, replay.hip);
//                       gradients w.r.t. the online variables only, each clipped by its OWN norm (tf.clip_by_norm, 10)
//   act               deepq/build_graph.py:146-199: argmax_a q, replaced by a uniform random action with probability eps
#pragma once
#include "qheads.hip.h"

struct mrl_qnet {
    mrl_qnet_desc qd;
    mrl_model base;                 // tensor table / P / ob_elems; base.pi = the feature network
    Net av, sv;                     // action_value / state_value heads (fc layers on the flattened latent)
    int nlat, lat_act;
    bool dueling;
    std::vector<int> init_kind;     // per tensor: 0 zeros, 1 orthogonal (scale in base.tensors), 2 xavier uniform
    // learner step (round 6): the target network's forward pass runs on a side stream next to the online network's (every kernel of a
    // batch-32 step is latency-bound on a handful of workgroups); created on the first eager mrl_qnet_td_grad call
    mutable hipStream_t side = nullptr, side2 = nullptr, side3 = nullptr;     // side2 / side3: weight gradients of consecutive layers
    mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    mutable bool side_failed = false;
    ~mrl_qnet() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
        if (side2) (void)hipStreamDestroy(side2);
        if (side3) (void)hipStreamDestroy(side3);
    }
};

static long q_add_tensor(mrl_qnet* q, const std::string& name, std::vector<int> shape, int kind, double scale) {
    long off = add_tensor(&q->base, name, shape, scale);
    q->init_kind.push_back(kind);
    return off;
}

static int q_build_heads(mrl_qnet* q, Net& net, const std::string& scope, int nout) {
    int nin = q->nlat;
    for (int i = 0; i <= q->qd.nhidden; ++i) {
        const bool last = i == q->qd.nhidden;
        Layer f{};
        f.kind = 1; f.K = nin; f.N = last ? nout : q->qd.hiddens[i]; f.act = last ? ACT_NONE : ACT_RELU;
        if (f.N < 1) return MRL_EINVAL;
        // tf.contrib.layers.fully_connected names: fully_connected, fully_connected_1, ... inside the scope
        std::string nm = scope + (i ? "/fully_connected_" + std::to_string(i) : std::string("/fully_connected"));
        f.w_off = q_add_tensor(q, nm + "/weights", {f.K, f.N}, 2, 1.0);
        f.b_off = q_add_tensor(q, nm + "/biases", {f.N}, 0, -1.0);
        if (q->qd.layer_norm && !last) {
            // deepq/models.py:27-29, 37-39: layers.layer_norm(out, center=True, scale=True) between the affine map and the
            // ReLU of every hidden head layer -> LayerNorm/beta, LayerNorm/gamma (LayerNorm_1/... for the next) in the scope
            std::string ln = scope + (i ? "/LayerNorm_" + std::to_string(i) : std::string("/LayerNorm"));
            f.ln = true;
            f.beta_off = q_add_tensor(q, ln + "/beta", {f.N}, 0, -1.0);
            f.gamma_off = q_add_tensor(q, ln + "/gamma", {f.N}, 3, -2.0);
        }
        f.out_elems = f.N;
        snprintf(f.name, sizeof f.name, "%s%d", scope.find("action") != std::string::npos ? "av" : "sv", i);
        net.L.push_back(f);
        nin = f.N;
    }
    net.nlat = nin; net.lat_act = ACT_NONE;
    return 0;
}

extern "C" int mrl_qnet_create(const mrl_qnet_desc* d, mrl_qnet** out) {
    if (!d || !out || d->nact < 1 || d->ob_ndim < 1 || d->ob_ndim > 3 || d->nhidden < 0 || d->nhidden > 4) return MRL_EINVAL;
    mrl_qnet* q = new mrl_qnet();
    q->qd = *d;
    q->dueling = d->dueling != 0;
    mrl_model& m = q->base;
    m.P = 0; m.ob_elems = 1;
    for (int i = 0; i < d->ob_ndim; ++i) m.ob_elems *= d->ob_shape[i];
    memset(&m.d, 0, sizeof m.d);
    m.d.network = d->network; m.d.ob_ndim = d->ob_ndim; m.d.ob_dtype = d->ob_dtype;
    for (int i = 0; i < 3; ++i) m.d.ob_shape[i] = d->ob_shape[i];
    m.d.num_layers = d->num_layers; m.d.num_hidden = d->num_hidden; m.d.activation = d->activation;
    m.d.nact = d->nact; m.d.pd_kind = MRL_PD_CATEGORICAL;
    m.d.layer_norm = (d->network == MRL_NET_MLP && d->body_layer_norm) ? 1 : 0;       // mlp(layer_norm=True) as the Q body
    m.vf_copy = false; m.has_pi_head = false; m.HP = 0;
    const std::string scope = "deepq/q_func";
    int rc = 0;
    if (d->network == MRL_NET_CONV_ONLY) {
        if (d->ob_ndim != 3 || d->ob_dtype != MRL_OB_U8 || d->ob_shape[2] % 4 != 0 || d->nconv < 1 || d->nconv > 4) rc = MRL_EUNSUP;
        int H = d->ob_shape[0], W = d->ob_shape[1], C = d->ob_shape[2];
        for (int i = 0; i < d->nconv && !rc; ++i) {
            const int nf = d->convs[i][0], rf = d->convs[i][1], st = d->convs[i][2];
            if (nf < 1 || nf % 4 != 0 || rf < 1 || st < 1) { rc = MRL_EUNSUP; break; }
            Layer l{};
            l.kind = 0; l.H = H; l.W = W; l.C = C; l.rf = rf; l.stride = st; l.NF = nf;
            // SAME: out = ceil(in / stride); total padding = max((out - 1)*stride + rf - in, 0), the smaller half in front
            l.OH = (H + st - 1) / st; l.OW = (W + st - 1) / st;
            l.pad_t = std::max((l.OH - 1) * st + rf - H, 0) / 2;
            l.pad_l = std::max((l.OW - 1) * st + rf - W, 0) / 2;
            l.K = rf * rf * C; l.N = nf; l.act = ACT_RELU;
            std::string nm = scope + "/convnet/" + (i ? "Conv_" + std::to_string(i) : std::string("Conv"));
            l.w_off = q_add_tensor(q, nm + "/weights", {rf, rf, C, nf}, 2, 1.0);
            l.b_off = q_add_tensor(q, nm + "/biases", {nf}, 0, -1.0);
            l.out_elems = (long)l.OH * l.OW * nf;
            snprintf(l.name, sizeof l.name, "qc%d", i + 1);
            m.pi.L.push_back(l);
            H = l.OH; W = l.OW; C = nf;
        }
        m.pi.nlat = H * W * C; m.pi.lat_act = ACT_RELU;
    } else if (d->network == MRL_NET_MLP || d->network == MRL_NET_NATURE_CNN) {
        const size_t before = m.tensors.size();
        rc = build_net(&m, m.pi, scope);                                   // ortho_init weights (a2c/utils.py:20-35)
        for (size_t i = before; i < m.tensors.size(); ++i) q->init_kind.push_back(m.tensors[i].scale == -2.0 ? 3 : m.tensors[i].scale < 0 ? 0 : 1);   // -2: LayerNorm gamma (ones)
    } else {
        rc = MRL_EUNSUP;
    }
    if (rc) { delete q; return rc; }
    q->nlat = m.pi.nlat; q->lat_act = m.pi.lat_act;
    rc = q_build_heads(q, q->av, scope + "/action_value", d->nact);
    if (!rc && q->dueling) rc = q_build_heads(q, q->sv, scope + "/state_value", 1);
    if (rc) { delete q; return rc; }
    *out = q;
    return 0;
}
extern "C" void mrl_qnet_destroy(mrl_qnet* q) { delete q; }
extern "C" long mrl_qnet_num_params(const mrl_qnet* q) { return q ? q->base.P : 0; }
extern "C" int mrl_qnet_num_tensors(const mrl_qnet* q) { return q ? (int)q->base.tensors.size() : 0; }
extern "C" int mrl_qnet_tensor_info(const mrl_qnet* q, int i, char* name, int name_cap, int* ndim, int shape[4], long* offset,
                                    int* init_kind, double* init_scale) {
    if (!q || i < 0 || i >= (int)q->base.tensors.size()) return MRL_EINVAL;
    if (init_kind) *init_kind = q->init_kind[i];
    return mrl_model_tensor_info(&q->base, i, name, name_cap, ndim, shape, offset, init_scale);
}

// ---- workspace ------------------------------------------------------------------------------------------------------
struct QWs {
    NetWs feat, av, sv;
    float *q_t, *q_tp1, *q_tp1_on, *dq, *dlat_tmp;
    float* part; size_t part_floats;
    float *part2, *part3;    // split-K scratch of the weight gradients on the second / third side stream
    void* td_scratch;
    double* sqpart;          // per-tensor sum-of-squares partials
    float* zeros;
    size_t total;
};
// sum-of-squares partials of the per-variable clip: one per Q_CHUNK-element chunk of every tensor (q_sumsq_kernel)
static size_t q_sq_parts(const mrl_qnet* q) { return (size_t)(q->base.P / 4096 + (long)q->base.tensors.size() + 1); }

static void q_carve(const mrl_qnet* q, int B, char* base, QWs& ws) {
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
    size_t part_floats = 1;
    auto do_net = [&](const Net& net, NetWs& nw) {
        nw.h.clear(); nw.dz.clear(); nw.xhat.clear(); nw.istd.clear();
        for (const Layer& l : net.L) {
            nw.xhat.push_back(l.ln ? (float*)take((size_t)B * l.out_elems * 4) : nullptr);
            nw.istd.push_back(l.ln ? (float*)take((size_t)B * 4) : nullptr);
            if (l.ln) part_floats = std::max(part_floats, (size_t)LN_MAXBLK * 2 * l.N);
            nw.h.push_back((float*)take((size_t)B * l.out_elems * 4));
            nw.dz.push_back((float*)take((size_t)B * l.out_elems * 4));
            part_floats = std::max(part_floats, (size_t)max_split_floats(l.K, l.N, layer_rows(l, B)));
        }
        nw.planes = nullptr; nw.dbg = nullptr;
    };
    do_net(q->base.pi, ws.feat);
    do_net(q->av, ws.av);
    if (q->dueling) do_net(q->sv, ws.sv);
    if (q->qd.nhidden == 1)      // K-split partials of the fused heads' hidden layer (qheads.hip.h): up to 16 x [B][N_a + N_s]
        part_floats = std::max(part_floats, (size_t)16 * std::min(B, 256) * (q->av.L[0].N + (q->dueling ? q->sv.L[0].N : 0)));
    const size_t qa = (size_t)B * q->qd.nact * 4;
    ws.q_t = (float*)take(qa); ws.q_tp1 = (float*)take(qa); ws.q_tp1_on = (float*)take(qa); ws.dq = (float*)take(qa);
    ws.dlat_tmp = (float*)take((size_t)B * q->nlat * 4);
    ws.part = (float*)take(part_floats * 4);
    ws.part2 = (float*)take(part_floats * 4);
    ws.part3 = (float*)take(part_floats * 4);
    ws.part_floats = part_floats;
    ws.td_scratch = take(mrl_dqn_td_scratch_bytes(B));
    ws.sqpart = (double*)take(q_sq_parts(q) * 8);
    ws.zeros = (float*)take(2048);
    ws.total = off;
}
extern "C" size_t mrl_qnet_workspace_bytes(const mrl_qnet* q, int batch) {
    if (!q || batch <= 0) return 0;
    QWs ws;
    q_carve(q, batch, nullptr, ws);
    return ws.total;
}

// ---- small kernels --------------------------------------------------------------------------------------------------
// q = V + (A - mean_a A)  (deepq/models.py:37-40); sv == nullptr: q = A
__global__ __launch_bounds__(256) void q_dueling_fwd_kernel(const float* __restrict__ av, const float* __restrict__ sv,
                                                            float* __restrict__ q, int B, int nA) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* a = av + (long)b * nA;
    float mean = 0.f;
    if (sv) {
        float s = 0.f;
        for (int j = 0; j < nA; ++j) s += a[j];
        mean = s / (float)nA;
    }
    for (int j = 0; j < nA; ++j) q[(long)b * nA + j] = sv ? sv[b] + (a[j] - mean) : a[j];
}
// dA = dq - mean_a dq;  dV = sum_a dq
__global__ __launch_bounds__(256) void q_dueling_bwd_kernel(const float* __restrict__ dq, float* __restrict__ dav,
                                                            float* __restrict__ dsv, int B, int nA) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* d = dq + (long)b * nA;
    float s = 0.f;
    for (int j = 0; j < nA; ++j) s += d[j];
    const float mean = s / (float)nA;
    for (int j = 0; j < nA; ++j) dav[(long)b * nA + j] = dsv ? d[j] - mean : d[j];
    if (dsv) dsv[b] = s;
}
// eps-greedy (build_graph.py:181-190): argmax_a q (first maximum wins, tf.argmax), a random action where u < eps
__global__ __launch_bounds__(256) void q_act_kernel(const float* __restrict__ q, const float* __restrict__ u,
                                                    const int32_t* __restrict__ rnd, float eps, int32_t* __restrict__ act,
                                                    int B, int nA) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* r = q + (long)b * nA;
    int best = 0;
    float bv = r[0];
    for (int j = 1; j < nA; ++j)
        if (r[j] > bv) { bv = r[j]; best = j; }
    act[b] = (u && u[b] < eps) ? rnd[b] : best;
}
// mean over the batch of KL(softmax(qa) || softmax(qb)) (build_graph.py:283-284: the distance between the greedy policies of
// the unperturbed and the adaptively perturbed network that steers the parameter-noise scale); one workgroup, fixed order
__global__ __launch_bounds__(256) void q_policy_kl_kernel(const float* __restrict__ qa, const float* __restrict__ qb, int B, int nA,
                                                          float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* a = qa + (long)b * nA;
        const float* c = qb + (long)b * nA;
        float ma = a[0], mb = c[0];
        for (int j = 1; j < nA; ++j) { ma = fmaxf(ma, a[j]); mb = fmaxf(mb, c[j]); }
        float za = 0.f, zb = 0.f;
        for (int j = 0; j < nA; ++j) { za += expf(a[j] - ma); zb += expf(c[j] - mb); }
        const float la = logf(za), lb = logf(zb);
        float kl = 0.f;
        for (int j = 0; j < nA; ++j) {
            const float lpa = (a[j] - ma) - la, lpb = (c[j] - mb) - lb;      // log softmax
            kl += expf(lpa) * (lpa - lpb);
        }
        s += (double)kl;
    }
    const double t = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[0] = (float)(t / (double)B);
}
// Round 6: the data gradient of the head(s) into the latent at learner batch sizes.  dlat[b][k] = act'(lat[b][k]) * (sum_n dzA[b][n] WA[k][n]
// + sum_n dzB[b][n] WB[k][n]) is a 32-row GEMM (batch 32: deepq.py:100) whose 2 x 7744 x 256 weights are read once: on the 128 x 128
// tile engine it ran as 2 launches of 61 workgroups with 32 of 128 rows live and 8 dependent k steps each (48 + 68 us of a 730 us
// learner step).  Here a workgroup owns 32 latent columns, its NW waves split the reduction (both heads' hidden units, concatenated)
// and run it on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) straight from 16-byte loads --
// lane (r, half) holds dz[b0 + r][8c + 4 half ..+3] and W[k0 + r][the same four n]: the reduction index of an MFMA step only has to
// be the SAME n on both operands --, the partial tiles are summed through LDS in wave order (deterministic).
struct QLatDgradArgs {
    const float* dz[2]; const float* W[2]; int N[2];     // per head: dz [B][N], W [K][N]; N % 8 == 0 (N[1] == 0: one head)
    const float* h; int act;                              // latent activations (act' source) or nullptr
    float* out; int B, K;
    int w16[2];                                           // W rows 16-byte aligned (else 8: two 8-byte loads per fragment)
};
__device__ __forceinline__ float4 q_ld4(const float* p, bool a16) {
    if (a16) return *reinterpret_cast<const float4*>(p);
    const float2 lo = *reinterpret_cast<const float2*>(p), hi = *reinterpret_cast<const float2*>(p + 2);
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
template <int NW, bool W16>
__global__ __launch_bounds__(64 * NW) void q_lat_dgrad_kernel(QLatDgradArgs a) {
    __shared__ float red[NW][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hf = lane >> 5;
    const int k0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int c0 = a.N[0] >> 3, ctot = c0 + (a.N[1] >> 3);
    const int per = (ctot + NW - 1) / NW, cb = wave * per, ce = min(ctot, cb + per);
    const bool brow = b0 + r < a.B, krow = k0 + r < a.K;
    const float* dzr[2] = {a.dz[0] + (long)(b0 + r) * a.N[0] + 4 * hf, a.dz[1] ? a.dz[1] + (long)(b0 + r) * a.N[1] + 4 * hf : nullptr};
    const float* wr[2] = {a.W[0] + (long)(k0 + r) * a.N[0] + 4 * hf, a.W[1] ? a.W[1] + (long)(k0 + r) * a.N[1] + 4 * hf : nullptr};
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // chunks in groups of U, the next group's loads issued before the current group's MFMAs (two register sets, ping-pong)
    constexpr int U = 4;
    float4 fa[2][U], fb[2][U];
    auto load = [&](int set, int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + u;
            const bool live = cc < ce;
            const int hd = cc >= c0;
            const long o = 8L * (hd ? cc - c0 : cc);
            fa[set][u] = mrl::ldz4(dzr[hd] + o, a.dz[0], live && brow);              // branch-free (convskinny.hip.h)
            fb[set][u] = mrl::ldz4w<W16>(wr[hd] + o, a.W[0], live && krow);
        }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].x, fb[set][u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].y, fb[set][u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].z, fb[set][u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u].w, fb[set][u].w, acc, 0, 0, 0);
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
    for (int i = 0; i < 16; ++i) red[wave][(i >> 2) * 8 + hf * 4 + (i & 3)][r] = acc[i];      // row = sample, column = latent unit
    __syncthreads();
    for (int idx = threadIdx.x; idx < 1024; idx += 64 * NW) {
        const int b = idx >> 5, k = idx & 31;
        if (b0 + b >= a.B || k0 + k >= a.K) continue;
        float v = red[0][b][k];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][b][k];
        const long o = (long)(b0 + b) * a.K + k0 + k;
        a.out[o] = a.h ? v * act_bwd_from_out(a.h[o], a.act) : v;
    }
}
// out = (out_prev + acc) * act'(h): second half of a two-source data gradient
struct EpiAddMaskAct {
    static constexpr bool HAS_BIAS = false;
    float* out; long ld; const float* add; const float* h; int act;
    __device__ __forceinline__ long addr(int m, int n, int) const { return (long)m * ld + n; }
    __device__ __forceinline__ float aux(long o, int) const { return add ? add[o] : 0.f; }
    __device__ __forceinline__ void put(long o, float acc, float a) const {
        const float v = acc + a;
        out[o] = h ? v * act_bwd_from_out(h[o], act) : v;
    }
};

// Round 6: the per-variable clip + Adam pair walked every tensor with a fixed 64 blocks of scalar accesses -- the two 7744 x 256 head
// matrices of the dueling conv_only net are 95 % of the parameters, so 128 of the ~900 blocks did nearly all the work in 121
// dependent iterations each (q_adam 60 us + q_sumsq 37 us of a 830 us learner step).  Now a block owns one CHUNK of Q_CHUNK
// consecutive elements of one tensor (16-byte accesses; tensors start 16-byte aligned or fall back to scalars): the table maps
// chunk -> tensor by prefix sums, big tensors get many blocks, small ones one.
constexpr int Q_CHUNK = 4096;
struct QTensorTable { int n; long off[40]; long size[40]; int cbeg[41]; };      // cbeg[t]: first chunk of tensor t; cbeg[n]: chunks in all
__device__ __forceinline__ int q_chunk_tensor(const QTensorTable& tt, int c) {
    int t = 0;
    while (t + 1 < tt.n && c >= tt.cbeg[t + 1]) ++t;
    return t;
}
__global__ __launch_bounds__(256) void q_sumsq_kernel(const float* __restrict__ g, QTensorTable tt, double* __restrict__ part) {
    __shared__ double sh[4];
    const int c = blockIdx.x, t = q_chunk_tensor(tt, c);
    const long lo = (long)(c - tt.cbeg[t]) * Q_CHUNK, hi = min(tt.size[t], lo + Q_CHUNK);
    const float* p = g + tt.off[t];
    double s = 0.0;
    if ((tt.off[t] & 3) == 0) {
        const long lo4 = lo, n4 = (hi - lo) / 4;
        for (long i = threadIdx.x; i < n4; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(p + lo4 + 4 * i);
            s += (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
        }
        for (long i = lo + 4 * n4 + threadIdx.x; i < hi; i += 256) s += (double)p[i] * (double)p[i];
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) s += (double)p[i] * (double)p[i];
    }
    const double r = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[c] = r;
}
// per-variable tf.clip_by_norm (build_graph.py:416-421: t * clip / max(||t||, clip)) then TF-1 ApplyAdam
__global__ __launch_bounds__(256) void q_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, QTensorTable tt, const double* __restrict__ part,
                                                     float alpha, const float* __restrict__ alpha_dev, float beta1,
                                                     float beta2, float eps, float clip) {
    __shared__ double sh[4];
    __shared__ float s_scale;
    const int c = blockIdx.x, t = q_chunk_tensor(tt, c);
    if (alpha_dev) alpha = alpha_dev[0];      // step size kept in device memory (replayable launch graphs)
    float scale = 1.f;
    if (clip > 0.f) {                         // the tensor's chunk partials in a fixed order: every block of the tensor forms the same sum
        double s = 0.0;
        for (int i = tt.cbeg[t] + (int)threadIdx.x; i < tt.cbeg[t + 1]; i += 256) s += part[i];
        const double tot = block_sum_256(s, sh);
        if (threadIdx.x == 0) {
            const float nrm = (float)sqrt(tot);
            s_scale = clip / fmaxf(nrm, clip);
        }
        __syncthreads();
        scale = s_scale;
    }
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    const long o = tt.off[t];
    const long lo = (long)(c - tt.cbeg[t]) * Q_CHUNK, hi = min(tt.size[t], lo + Q_CHUNK);
    auto upd = [&](float gi, float& mi, float& vi, float& pi) {
        const float x = gi * scale;
        mi = mi + (x - mi) * omb1;
        vi = vi + (x * x - vi) * omb2;
        pi = pi - (mi * alpha) / (sqrtf(vi) + eps);
        return x;
    };
    long i0 = lo;
    if ((o & 3) == 0) {
        const long n4 = (hi - lo) / 4;
        for (long i = threadIdx.x; i < n4; i += 256) {
            const long e = o + lo + 4 * i;
            float4 g4 = *reinterpret_cast<float4*>(g + e), m4 = *reinterpret_cast<float4*>(m + e), v4 = *reinterpret_cast<float4*>(v + e),
                   p4 = *reinterpret_cast<float4*>(p + e);
            g4.x = upd(g4.x, m4.x, v4.x, p4.x); g4.y = upd(g4.y, m4.y, v4.y, p4.y);
            g4.z = upd(g4.z, m4.z, v4.z, p4.z); g4.w = upd(g4.w, m4.w, v4.w, p4.w);
            *reinterpret_cast<float4*>(g + e) = g4; *reinterpret_cast<float4*>(m + e) = m4;
            *reinterpret_cast<float4*>(v + e) = v4; *reinterpret_cast<float4*>(p + e) = p4;
        }
        i0 = lo + 4 * n4;
    }
    for (long i = i0 + threadIdx.x; i < hi; i += 256) {
        float mi = m[o + i], vi = v[o + i], pi = p[o + i];
        g[o + i] = upd(g[o + i], mi, vi, pi);
        m[o + i] = mi; v[o + i] = vi; p[o + i] = pi;
    }
}

// ---- fused heads (qheads.hip.h) ---------------------------------------------------------------------------------------
static int q_lat_dgrad(const mrl_qnet* q, const float* params, QWs& ws, float* dlat, int B, hipStream_t st);
static bool q_heads_fused_ok(const mrl_qnet* q, int B) {
    if (B > 256 || q->qd.nhidden != 1 || q->qd.layer_norm || q->qd.nact > QH_MAXOUT || q->nlat % 8 || q->base.pi.L.empty() ||
        !get_option("dqn_heads", "MRL_DQN_HEADS", 1))
        return false;
    if (!tune_table().empty())
        for (const char* nm : {"av0", "av1", "sv0", "sv1"})
            for (const char* pass : {".fwd", ".wgrad", ".dgrad"})
                if (tune_table().count(std::string(nm) + pass)) return false;
    if (q->av.L.size() != 2 || q->av.L[0].N % 32 || (q->dueling && (q->sv.L.size() != 2 || q->sv.L[0].N % 32))) return false;
    return true;
}
static QHeads q_heads_desc(const mrl_qnet* q, const float* params) {
    QHeads hd{};
    const Net* nets[2] = {&q->av, &q->sv};
    hd.nheads = q->dueling ? 2 : 1;
    hd.K = q->nlat;
    for (int i = 0; i < hd.nheads; ++i) {
        const Layer &l0 = nets[i]->L[0], &l1 = nets[i]->L[1];
        hd.W0[i] = params + l0.w_off; hd.b0[i] = params + l0.b_off; hd.W1[i] = params + l1.w_off; hd.b1[i] = params + l1.b_off;
        hd.N0[i] = l0.N; hd.nout[i] = l1.N;
    }
    return hd;
}
// latent [B][K] -> hidden activations (ws.av.h[0], ws.sv.h[0]), raw head outputs (h[1]) and q_out; with a second problem (other
// parameters, latent, batch, workspace: the target pass next to the online pass) both go through the same two launches
struct QPass { const float* params; const void* obs; int B; QWs* ws; float* q_out; };
static int q_heads_forward_fused(const mrl_qnet* q, const float* lat, const float* params, QWs& ws, int B, float* q_out, hipStream_t st,
                                 const QPass* second = nullptr) {
    const QHeads hd = q_heads_desc(q, params);
    const int ntot = hd.N0[0] + (hd.nheads > 1 ? hd.N0[1] : 0), ntiles = ntot / 32, kc = hd.K / 8;
    const int MT = (B <= 32 || second) ? 1 : 2, zg = (B + 32 * MT - 1) / (32 * MT);
    const int zg1 = second ? (second->B + 31) / 32 : 0;
    const int Bmax = second ? std::max(B, second->B) : B;
    long S = std::max(1, std::min(16, 512 / (ntiles * (zg + zg1))));
    S = std::min<long>(S, (long)(ws.part_floats / ((size_t)Bmax * ntot)));
    if (second) S = std::min<long>(S, (long)(second->ws->part_floats / ((size_t)Bmax * ntot)));
    S = std::min<long>(S, std::max(1, kc / 8));
    if (S < 1) return MRL_ENOSPC;
    const int cps = (int)((kc + S - 1) / S);
    S = (kc + cps - 1) / cps;
    const QHeadsFwdProb f0{hd, lat, B, ws.part};
    QHeadsFwdProb f1 = f0;
    QHeadsOutProb o0{hd, ws.part, (int)S, B, ws.av.h[0], q->dueling ? ws.sv.h[0] : nullptr, ws.av.h[1], q->dueling ? ws.sv.h[1] : nullptr, q_out};
    QHeadsOutProb o1 = o0;
    if (second) {
        QWs& w2 = *second->ws;
        const QHeads hd2 = q_heads_desc(q, second->params);
        f1 = QHeadsFwdProb{hd2, w2.feat.h.back(), second->B, w2.part};
        o1 = QHeadsOutProb{hd2, w2.part, (int)S, second->B, w2.av.h[0], q->dueling ? w2.sv.h[0] : nullptr, w2.av.h[1],
                           q->dueling ? w2.sv.h[1] : nullptr, second->q_out};
    }
    {
        ProfScope ps("heads.fwd", 2.0 * (B + (second ? second->B : 0)) * (double)hd.K * ntot, 0.0, st);
        if (MT == 1) hipLaunchKernelGGL(q_heads_fwd_kernel<1>, dim3(ntiles, (int)S, zg + zg1), dim3(512), 0, st, f0, f1, zg, cps);
        else hipLaunchKernelGGL(q_heads_fwd_kernel<2>, dim3(ntiles, (int)S, zg), dim3(512), 0, st, f0, f1, zg, cps);
        MRL_LAUNCH_CHECK();
    }
    ProfScope ps("heads.out", 0.0, 4.0 * (B + (second ? second->B : 0)) * (double)ntot * (S + 2), st);
    hipLaunchKernelGGL(q_heads_out_kernel, dim3(B + (second ? second->B : 0)), dim3(256), 0, st, o0, o1);
    MRL_LAUNCH_CHECK();
    return 0;
}
// The online and the target pass of a learner step as ONE sequence of launches (each kernel takes both problems): possible when every
// body layer is a conv layer the skinny-tile kernel takes and the heads have the fused form.  false: not this network / these sizes.
static bool q_forward_pair_ok(const mrl_qnet* q, const QPass& a, const QPass& b) {
    if (!get_option("dqn_pair", "MRL_DQN_PAIR", 1) || !q_heads_fused_ok(q, a.B) || !q_heads_fused_ok(q, b.B)) return false;
    const bool u8 = q->qd.ob_dtype == MRL_OB_U8;
    for (size_t i = 0; i < q->base.pi.L.size(); ++i) {
        const Layer& l = q->base.pi.L[i];
        if (l.kind != 0 || l.ln) return false;
        for (const QPass* p : {&a, &b}) {
            const void* src = i ? (const void*)p->ws->feat.h[i - 1] : p->obs;
            if (!conv_skinny_ok(l, p->B * l.OH * l.OW, src, "fwd") || (!(i == 0 && !u8) && wres_fwd_ok(l, i == 0 && u8, src))) return false;
        }
    }
    return true;
}
static int q_forward_pair(const mrl_qnet* q, const QPass& a, const QPass& b, hipStream_t st) {
    const bool u8 = q->qd.ob_dtype == MRL_OB_U8;
    for (size_t i = 0; i < q->base.pi.L.size(); ++i) {
        const Layer& l = q->base.pi.L[i];
        ConvGeom ga, gb;
        fill_conv(ga, l, i ? (const void*)a.ws->feat.h[i - 1] : a.obs, a.B * l.OH * l.OW, nullptr);
        fill_conv(gb, l, i ? (const void*)b.ws->feat.h[i - 1] : b.obs, b.B * l.OH * l.OW, nullptr);
        const ConvSkinnyProb pb{gb, b.params + l.w_off, b.params + l.b_off, b.ws->feat.h[i]};
        char label[40];
        if (prof_enabled()) snprintf(label, sizeof label, "%s.fwd", l.name);
        ProfScope ps(label, 2.0 * (ga.npix + gb.npix) * (double)l.K * l.NF, 0.0, st);
        const float *W = a.params + l.w_off, *bias = a.params + l.b_off;
        hipError_t e = i ? launch_conv_skinny_fwd<0>(ga, W, bias, a.ws->feat.h[i], l.NF, l.act, st, &pb)
                         : u8 ? launch_conv_skinny_fwd<1>(ga, W, bias, a.ws->feat.h[i], l.NF, l.act, st, &pb)
                              : launch_conv_skinny_fwd<2>(ga, W, bias, a.ws->feat.h[i], l.NF, l.act, st, &pb);
        if (e != hipSuccess) return (int)e;
    }
    return q_heads_forward_fused(q, a.ws->feat.h.back(), a.params, *a.ws, a.B, a.q_out, st, &b);
}
// dq [B][nA] -> gradients of all four head tensors (+ biases) in the flat gradient, and (q_lat_dgrad) the latent's gradient
static int q_heads_backward_fused(const mrl_qnet* q, const float* lat, const float* params, QWs& ws, float* grads, float* dlat, int B,
                                  hipStream_t st, StepCtx& ctx) {
    const QHeads hd = q_heads_desc(q, params);
    const Layer &a0 = q->av.L[0], &a1 = q->av.L[1];
    const Layer* s0 = q->dueling ? &q->sv.L[0] : nullptr;
    const Layer* s1 = q->dueling ? &q->sv.L[1] : nullptr;
    const int ntot = hd.N0[0] + (hd.nheads > 1 ? hd.N0[1] : 0);
    {
        ProfScope ps("heads.bwd", 0.0, 12.0 * B * (double)ntot, st);
        hipLaunchKernelGGL(q_heads_bwd_kernel, dim3(B + hd.nheads), dim3(256), (size_t)B * hd.nout[0] * sizeof(float), st, hd, ws.dq, B,
                           ws.av.h[0], s0 ? ws.sv.h[0] : nullptr, ws.av.dz[1], s0 ? ws.sv.dz[1] : nullptr, ws.av.dz[0],
                           s0 ? ws.sv.dz[0] : nullptr, grads + a1.w_off, grads + a1.b_off, s1 ? grads + s1->w_off : nullptr,
                           s1 ? grads + s1->b_off : nullptr);
        MRL_LAUNCH_CHECK();
    }
    hipStream_t stw = st;
    if (ctx.nwstream) {
        const int j = ctx.wrr++ % ctx.nwstream;
        MRL_HIP_CHECK(hipEventRecord(ctx.ev_fork, st));
        MRL_HIP_CHECK(hipStreamWaitEvent(ctx.wstream[j], ctx.ev_fork, 0));
        stw = ctx.wstream[j];
    }
    {
        ProfScope ps("heads0.wgrad", 2.0 * B * (double)hd.K * ntot, 0.0, stw);
        hipLaunchKernelGGL(q_heads_wgrad_kernel, dim3((hd.K + 31) / 32, (ntot / 32 + 3) / 4), dim3(256), 0, stw, hd, lat, ws.av.dz[0],
                           s0 ? ws.sv.dz[0] : nullptr, B, grads + a0.w_off, grads + a0.b_off, s0 ? grads + s0->w_off : nullptr,
                           s0 ? grads + s0->b_off : nullptr);
        MRL_LAUNCH_CHECK();
    }
    return q_lat_dgrad(q, params, ws, dlat, B, st);
}

// ---- forward --------------------------------------------------------------------------------------------------------
static int q_heads_forward(const mrl_qnet* q, const Net& net, const float* lat, const float* params, NetWs& nw, int B,
                           hipStream_t st, float* part, size_t part_floats) {
    In in{lat, nullptr};
    for (size_t i = 0; i < net.L.size(); ++i) {
        Layer lcopy;
        const Layer* lp = &net.L[i];
        if (lp->ln) { lcopy = *lp; lcopy.act = ACT_NONE; lp = &lcopy; }          // the affine map alone; LN + ReLU below
        int rc = layer_forward<kExp>(&q->base, *lp, i == 0, in, i ? nw.h[i - 1] : nullptr, params, nw.h[i], nullptr, nullptr, B, st,
                                     nullptr, nullptr, nullptr, nullptr, nullptr, part, part_floats);
        if (rc) return rc;
        if (net.L[i].ln) {
            const Layer& l = net.L[i];
            const int blocks = (int)std::min<long>(((long)B + 3) / 4, 2048);
            hipLaunchKernelGGL(ln_fwd_kernel, dim3(blocks), dim3(256), 0, st, nw.h[i], nw.xhat[i], nw.istd[i], params + l.beta_off,
                               params + l.gamma_off, B, l.N, l.act);
            MRL_LAUNCH_CHECK();
        }
    }
    return 0;
}
static int q_forward(const mrl_qnet* q, const float* params, const void* obs, int B, QWs& ws, float* q_out, hipStream_t st) {
    In in{obs, nullptr};
    int rc = net_forward(&q->base, q->base.pi, in, params, ws.feat, B, st, ws.part, ws.part_floats);
    if (rc) return rc;
    const float* lat = ws.feat.h.back();
    if (q_heads_fused_ok(q, B)) return q_heads_forward_fused(q, lat, params, ws, B, q_out, st);
    if ((rc = q_heads_forward(q, q->av, lat, params, ws.av, B, st, ws.part, ws.part_floats))) return rc;
    if (q->dueling && (rc = q_heads_forward(q, q->sv, lat, params, ws.sv, B, st, ws.part, ws.part_floats))) return rc;
    hipLaunchKernelGGL(q_dueling_fwd_kernel, dim3((B + 255) / 256), dim3(256), 0, st, ws.av.h.back(),
                       q->dueling ? ws.sv.h.back() : nullptr, q_out, B, q->qd.nact);
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrl_qnet_values(const mrl_qnet* q, const float* params, const void* obs, int n, float* q_out, void* workspace,
                               size_t workspace_bytes, int batch, void* stream) {
    if (!q || !params || !obs || !q_out || !workspace || n <= 0 || batch <= 0) return MRL_EINVAL;
    QWs ws;
    q_carve(q, batch, (char*)workspace, ws);
    if (ws.total > workspace_bytes) return MRL_ENOSPC;
    const size_t ob_bytes = (size_t)q->base.ob_elems * (q->qd.ob_dtype == MRL_OB_U8 ? 1 : 4);
    for (int c0 = 0; c0 < n; c0 += batch) {
        const int Bc = std::min(batch, n - c0);
        int rc = q_forward(q, params, (const char*)obs + (size_t)c0 * ob_bytes, Bc, ws, q_out + (size_t)c0 * q->qd.nact,
                           (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int mrl_qnet_act(const mrl_qnet* q, const float* params, const void* obs, int n, float eps, const float* uniforms,
                            const int32_t* rand_actions, int32_t* actions_out, float* q_out, void* workspace,
                            size_t workspace_bytes, int batch, void* stream) {
    if (!q || !actions_out || n > batch || (uniforms && !rand_actions)) return MRL_EINVAL;
    QWs ws;
    q_carve(q, batch, (char*)workspace, ws);
    if (ws.total > workspace_bytes) return MRL_ENOSPC;
    float* qo = q_out ? q_out : ws.q_t;
    int rc = mrl_qnet_values(q, params, obs, n, qo, workspace, workspace_bytes, batch, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(q_act_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, qo, uniforms, rand_actions, eps,
                       actions_out, n, q->qd.nact);
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrl_qnet_policy_kl(const float* q_a, const float* q_b, int n, int nact, float* mean_kl_out, void* stream) {
    if (!q_a || !q_b || !mean_kl_out || n <= 0 || nact <= 0) return MRL_EINVAL;
    hipLaunchKernelGGL(q_policy_kl_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q_a, q_b, n, nact, mean_kl_out);
    MRL_LAUNCH_CHECK();
    return 0;
}

// ---- TD gradient --- deepq/build_graph.py:380-421 -------------------------------------------------------------------
static int q_heads_backward(const mrl_qnet* q, const Net& net, const float* lat, const float* params, NetWs& nw, QWs& qws,
                            float* grads, float* dlat, bool add, int B, hipStream_t st, StepCtx& ctx, bool into_latent = true) {
    // weight gradients of the head layers + data gradients between them (net_backward treats `in.obs` as the fc input)
    Ws ws{};
    ws.part = qws.part; ws.part_floats = qws.part_floats; ws.zeros = qws.zeros;
    In in{lat, nullptr};
    int rc = net_backward<kExp>(&q->base, net, in, params, nw, ws, grads, B, 0, st, ctx, false);
    if (rc) return rc;
    if (!into_latent) return 0;
    // into the latent: dlat (+)= dz0 @ W0^T, masked by act'(latent) when it is the last contribution
    const Layer& l0 = net.L[0];
    const float* dz = nw.dz[0];
    const float* W = params + l0.w_off;
    RowKC af{dz, l0.N, B, l0.N, is_vec(dz, l0.N), nullptr};
    RowKC bf{W, l0.N, l0.K, l0.N, is_vec(W, l0.N), nullptr};
    const int dv = pick_variant(l0.name, "dgrad", B, l0.K);
    const float* hm = q->base.pi.L.empty() ? nullptr : qws.feat.h.back();
    if (add) {
        EpiAddMaskAct ef{dlat, l0.K, qws.dlat_tmp, hm, q->lat_act};
        return gemm_dispatch(l0.name, "dgrad", dv, af, bf, ef, B, l0.K, l0.N, 1, l0.N, st);
    }
    EpiMaskAct ef{q->dueling ? qws.dlat_tmp : dlat, l0.K, q->dueling ? nullptr : hm, q->lat_act};
    return gemm_dispatch(l0.name, "dgrad", dv, af, bf, ef, B, l0.K, l0.N, 1, l0.N, st);
}
// both heads' first layers into the latent in one launch (q_lat_dgrad_kernel); false: shapes / alignment it does not take
static bool q_lat_dgrad_ok(const mrl_qnet* q, const float* params, const QWs& ws, int B) {
    if (B > 256 || !get_option("dqn_latdgrad", "MRL_DQN_LATDGRAD", 1) || tune_table().count("av0.dgrad") || tune_table().count("sv0.dgrad"))
        return false;
    const Net* nets[2] = {&q->av, q->dueling ? &q->sv : nullptr};
    const NetWs* nws[2] = {&ws.av, &ws.sv};
    for (int i = 0; i < 2; ++i) {
        if (!nets[i]) continue;
        const Layer& l0 = nets[i]->L[0];
        if (l0.kind != 1 || l0.N % 8 || l0.K != nets[0]->L[0].K || (uintptr_t)(params + l0.w_off) % 8 || (uintptr_t)nws[i]->dz[0] % 16) return false;
    }
    return true;
}
static int q_lat_dgrad(const mrl_qnet* q, const float* params, QWs& ws, float* dlat, int B, hipStream_t st) {
    QLatDgradArgs a{};
    const Layer& la = q->av.L[0];
    a.dz[0] = ws.av.dz[0]; a.W[0] = params + la.w_off; a.N[0] = la.N; a.w16[0] = (uintptr_t)a.W[0] % 16 == 0;
    double fl = 2.0 * B * (double)la.K * la.N;
    if (q->dueling) {
        const Layer& ls = q->sv.L[0];
        a.dz[1] = ws.sv.dz[0]; a.W[1] = params + ls.w_off; a.N[1] = ls.N; a.w16[1] = (uintptr_t)a.W[1] % 16 == 0;
        fl += 2.0 * B * (double)ls.K * ls.N;
    }
    a.h = q->base.pi.L.empty() ? nullptr : ws.feat.h.back();
    a.act = q->lat_act; a.out = dlat; a.B = B; a.K = la.K;
    ProfScope ps("heads0.dgrad", fl, 0.0, st);
    if (a.w16[0] && (!q->dueling || a.w16[1])) hipLaunchKernelGGL((q_lat_dgrad_kernel<8, true>), dim3((a.K + 31) / 32, (B + 31) / 32), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((q_lat_dgrad_kernel<8, false>), dim3((a.K + 31) / 32, (B + 31) / 32), dim3(512), 0, st, a);
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrl_qnet_td_grad(const mrl_qnet* q, const float* params, const float* target_params, const void* obs_t,
                                const int32_t* act, const float* rew, const void* obs_tp1, const float* done,
                                const float* weights, float gamma, int double_q, int B, float* grads_out, float* td_out,
                                float* loss_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !params || !target_params || !obs_t || !act || !rew || !obs_tp1 || !done || !weights || !grads_out || !td_out ||
        !loss_out || !workspace || B <= 0)
        return MRL_EINVAL;
    if (q->base.pi.L.empty()) return MRL_EUNSUP;
    hipStream_t st = (hipStream_t)stream;
    const int nA = q->qd.nact;
    const size_t ob_bytes = (size_t)q->base.ob_elems * (q->qd.ob_dtype == MRL_OB_U8 ? 1 : 4);
    int rc;
    // Round 6.  (a) Double-Q evaluates the ONLINE network on obs_t and on obs_tp1 (build_graph.py:393-398): when the caller hands
    // the two batches over back to back, they go through the network as ONE batch of 2 B (rows 0 .. B-1 = obs_t, whose activations
    // feed the backward pass) -- one pass of latency-bound launches fewer.  (b) The TARGET network's pass is independent of both:
    // with room for a second workspace it runs on a side stream next to the online pass (fork / join with events; the caller's
    // stream -- and a graph captured from it -- sees one step).  Both need a workspace sized for it (qmodel.py allocates
    // mrl_qnet_workspace_bytes(2 B) + mrl_qnet_workspace_bytes(B)); otherwise the three passes run one after the other as before.
    const bool contiguous = (const char*)obs_tp1 == (const char*)obs_t + (size_t)B * ob_bytes;
    QWs ws, wt;
    q_carve(q, 2 * B, (char*)workspace, ws);
    q_carve(q, B, nullptr, wt);
    const bool roomy = ws.total + wt.total + 256 <= workspace_bytes && get_option("dqn_overlap", "MRL_DQN_OVERLAP", 1);
    const bool merged = roomy && double_q && contiguous;
    if (!roomy) {
        q_carve(q, B, (char*)workspace, ws);
        if (ws.total > workspace_bytes) return MRL_ENOSPC;
    } else {
        q_carve(q, B, (char*)workspace + ((ws.total + 255) & ~(size_t)255), wt);
    }
    bool par = roomy && !prof_enabled() && !q->side_failed;
    if (par && !q->side) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) par = false;      // not while capturing
        else {
            hipError_t e = hipStreamCreateWithFlags(&q->side, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&q->side2, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&q->side3, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&q->ev_fork, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&q->ev_join, hipEventDisableTiming);
            if (e != hipSuccess) { q->side_failed = true; par = false; (void)hipGetLastError(); }
        }
    }
    QWs& wtr = roomy ? wt : ws;
    const QPass pass_on{params, obs_t, 2 * B, &ws, ws.q_t}, pass_tg{target_params, obs_tp1, B, &wt, wt.q_tp1};
    const bool paired = merged && q_forward_pair_ok(q, pass_on, pass_tg);
    if (paired) {
        if ((rc = q_forward_pair(q, pass_on, pass_tg, st))) return rc;
    } else if (par) {
        MRL_HIP_CHECK(hipEventRecord(q->ev_fork, st));
        MRL_HIP_CHECK(hipStreamWaitEvent(q->side, q->ev_fork, 0));
        if ((rc = q_forward(q, target_params, obs_tp1, B, wtr, wtr.q_tp1, q->side))) return rc;
        MRL_HIP_CHECK(hipEventRecord(q->ev_join, q->side));
    } else {
        // the obs_tp1 passes first (their activations are not needed again), then obs_t whose activations feed the backward
        if ((rc = q_forward(q, target_params, obs_tp1, B, wtr, wtr.q_tp1, st))) return rc;
    }
    const float* q_tp1_on = nullptr;
    if (paired) {
        q_tp1_on = ws.q_t + (size_t)B * nA;
    } else if (merged) {
        if ((rc = q_forward(q, params, obs_t, 2 * B, ws, ws.q_t, st))) return rc;
        q_tp1_on = ws.q_t + (size_t)B * nA;
    } else {
        if (double_q) {
            if ((rc = q_forward(q, params, obs_tp1, B, ws, ws.q_tp1_on, st))) return rc;
            q_tp1_on = ws.q_tp1_on;
        }
        if ((rc = q_forward(q, params, obs_t, B, ws, ws.q_t, st))) return rc;
    }
    if (par && !paired) MRL_HIP_CHECK(hipStreamWaitEvent(st, q->ev_join, 0));
    if ((rc = mrl_dqn_td(ws.q_t, wtr.q_tp1, q_tp1_on, act, rew, done, weights, gamma, B, nA, td_out,
                         loss_out, ws.dq, ws.td_scratch, stream)))
        return rc;
    const float* lat = ws.feat.h.back();
    float* dlat = ws.feat.dz.back();
    const bool fused = q_heads_fused_ok(q, B) && q_lat_dgrad_ok(q, params, ws, B);
    if (!fused) {
        hipLaunchKernelGGL(q_dueling_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, st, ws.dq, ws.av.dz.back(),
                           q->dueling ? ws.sv.dz.back() : nullptr, B, nA);
        MRL_LAUNCH_CHECK();
    }
    // the weight gradients hang off the dz chain: with the side stream free again (the target pass was joined above) they run there,
    // next to the data gradients (StepCtx::wstream); joined below, before the caller's stream reads the gradient
    StepCtx ctx;
    if (par && get_option("dqn_wstream", "MRL_DQN_WSTREAM", 1)) {
        ctx.nwstream = std::min(3, get_option("dqn_wstream", "MRL_DQN_WSTREAM", 1) == 1 ? 3 : get_option("dqn_wstream", "MRL_DQN_WSTREAM", 1) - 1);
        ctx.wstream[0] = q->side; ctx.wstream[1] = q->side2; ctx.wstream[2] = q->side3;
        ctx.wpart[0] = ws.part; ctx.wpart[1] = ws.part2; ctx.wpart[2] = ws.part3;
        ctx.ev_fork = q->ev_fork; ctx.ev_join = q->ev_join;
    }
    const bool one_dgrad = q_lat_dgrad_ok(q, params, ws, B);
    if (fused) {
        if ((rc = q_heads_backward_fused(q, lat, params, ws, grads_out, dlat, B, st, ctx))) return rc;
    } else {
    if ((rc = q_heads_backward(q, q->av, lat, params, ws.av, ws, grads_out, dlat, false, B, st, ctx, !one_dgrad))) return rc;
    if (q->dueling && (rc = q_heads_backward(q, q->sv, lat, params, ws.sv, ws, grads_out, dlat, true, B, st, ctx, !one_dgrad))) return rc;
    if (one_dgrad && (rc = q_lat_dgrad(q, params, ws, dlat, B, st))) return rc;
    }
    Ws mws{};
    mws.part = ws.part; mws.part_floats = ws.part_floats; mws.zeros = ws.zeros;
    In in{obs_t, nullptr};
    rc = net_backward<kExp>(&q->base, q->base.pi, in, params, ws.feat, mws, grads_out, B, 0, st, ctx, false);
    for (int j = 0; j < ctx.nwstream; ++j) {
        MRL_HIP_CHECK(hipEventRecord(q->ev_join, ctx.wstream[j]));
        MRL_HIP_CHECK(hipStreamWaitEvent(st, q->ev_join, 0));
    }
    return rc;
}

// per-variable clip_by_norm + Adam (deepq/deepq.py:205-208: tf.train.AdamOptimizer(lr), grad_norm_clipping=10)
extern "C" int mrl_qnet_adam_step(const mrl_qnet* q, float* params, float* grads, float* adam_m, float* adam_v, float alpha,
                                  const float* alpha_dev, float beta1, float beta2, float eps, float grad_norm_clipping,
                                  void* workspace, size_t workspace_bytes, int batch, void* stream) {
    if (!q || !params || !grads || !adam_m || !adam_v || !workspace) return MRL_EINVAL;
    QWs ws;
    q_carve(q, batch, (char*)workspace, ws);
    if (ws.total > workspace_bytes) return MRL_ENOSPC;
    QTensorTable tt;
    tt.n = (int)q->base.tensors.size();
    if (tt.n > 40) return MRL_EUNSUP;
    int nchunks = 0;
    for (int i = 0; i < tt.n; ++i) {
        const TensorInfo& t = q->base.tensors[i];
        long n = 1;
        for (int k = 0; k < t.ndim; ++k) n *= t.shape[k];
        tt.off[i] = t.off; tt.size[i] = n;
        tt.cbeg[i] = nchunks;
        nchunks += (int)((n + Q_CHUNK - 1) / Q_CHUNK);
    }
    tt.cbeg[tt.n] = nchunks;
    if ((size_t)nchunks > q_sq_parts(q)) return MRL_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    if (grad_norm_clipping > 0.f) {
        hipLaunchKernelGGL(q_sumsq_kernel, dim3(nchunks), dim3(256), 0, st, grads, tt, ws.sqpart);
        MRL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(q_adam_kernel, dim3(nchunks), dim3(256), 0, st, params, grads, adam_m, adam_v, tt, ws.sqpart,
                       alpha, alpha_dev, beta1, beta2, eps, grad_norm_clipping);
    MRL_LAUNCH_CHECK();
    return 0;
}
