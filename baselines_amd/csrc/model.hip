// PPO2 policy/value model on gfx950: layout object, forward/backward orchestration over the
// MFMA implicit-GEMM template (gemm.hip.h), fused heads + loss + closed-form gradients,
// act-side sampling, deterministic reductions.
//
// Reference map (paths relative to baselines/):
//   networks   common/models.py:15-26 (nature_cnn), :74-103 (mlp); a2c/utils.py:37-63 (conv, fc)
//   heads      common/policies.py:43-64; common/distributions.py:59-113 (+ _matching_fc :351-355)
//   pd maths   common/distributions.py:153-204, 227-251
//   loss       ppo2/model.py:57-91;  adv normalisation ppo2/model.py:136-139
//   gather     ppo2/ppo2.py:157-165 (fused into the first-layer loaders; sf01 runner.py:69-74 eliminated)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <set>
#include <utility>
#include <string>
#include <vector>

#include "common.hip.h"
#include "gemm.hip.h"
#include "wres.hip.h"
#include "imgres.hip.h"
#include "ldsdgrad.hip.h"
#include "gemmx6.hip.h"
#include "dgradx6.hip.h"
#ifdef MRL_X6_EXPERIMENTS
#include "gemmx6s.hip.h"
#endif
#include "gemmx6r.hip.h"
#include "convx6c.hip.h"
#include "c1fwd.hip.h"
#include "wgradx8.hip.h"
#include "wgradtr.hip.h"
#include "c1wgrad.hip.h"
#include "mlpstep.hip.h"
#include "mlpact.hip.h"
#include "convskinny.hip.h"
#include "comm.hip.h"
#include "lstm.hip.h"

using namespace mrl;

// ============================================================================================
// host-side layout
// ============================================================================================
// Product builds carry the engines the options of include/mrl.h select.  -DMRL_X6_EXPERIMENTS (MRL_BUILD_DEFINES, csrc/build.py)
// adds the measured-and-dropped variants the A/B scripts and profiles/README.md refer to: the wave-specialised split engine
// (gemmx6s.hip.h), pre-split plane tensors (act_planes bits 1 / 2 / 32), phase-stamp and phase-omission knobs (*_dbg).
#ifdef MRL_X6_EXPERIMENTS
constexpr bool kExp = true;
#else
constexpr bool kExp = false;
#endif

struct Layer {
    int kind;   // 0 conv, 1 fc
    int H, W, C, rf, stride, OH, OW, NF;   // conv
    int pad_t, pad_l;                      // conv: zero padding above / left (SAME convolutions of `conv_only`; 0 = VALID)
    int K, N;                              // fc (and conv: K = rf*rf*C, N = NF)
    int act;
    long w_off, b_off;
    long out_elems;                        // per sample
    char name[16];                         // c1, c2, c3, fc1, mlp_fc0, ...
    bool ln = false;                       // layer normalisation between the affine map and the activation (mlp(layer_norm=True))
    long beta_off = -1, gamma_off = -1;    // [N] each, gamma right behind beta
};

struct Net {
    std::vector<Layer> L;                  // feature layers (conv / fc); empty for the plain `lstm` network
    int nlat;
    int lat_act;
    bool lstm = false;                     // recurrent cell on top of the features (a2c/utils.py:81-102)
    int lstm_nin = 0, nh = 0;
    long wx_off = -1, wh_off = -1, lb_off = -1;
    bool lnl = false;                      // layer-normalised cell (a2c/utils.py:110-140 lnlstm): gains / biases of the three LNs
    long gx_off = -1, gh_off = -1, gc_off = -1;       // each gain is followed by its bias (gx|bx, gh|bh, gc|bc)
};

struct TensorInfo {
    std::string name;
    int ndim;
    int shape[4];
    long off;
    double scale;  // <0: zeros
};

struct mrl_model {
    mrl_model_desc d;
    Net pi, vf;
    bool vf_copy;
    bool has_pi_head;
    long pi_w, pi_b, logstd, vf_w, vf_b;   // flat offsets (-1: absent)
    long head_off;                         // first head parameter (heads are contiguous at the tail)
    int HP;                                // number of head parameters
    long P;
    long ob_elems;
    std::vector<TensorInfo> tensors;
    mrl_comm* comm = nullptr;              // data-parallel communicator (mrl_model_attach_comm); not owned
    float rank_weight = 1.f;               // mpi_adam_optimizer.py:21 `flat_grad * mpi_rank_weight`
    const float* ext_advstat = nullptr;    // precomputed minibatch advantage statistics (mrl_model_set_advstat); not owned
};

static long add_tensor(mrl_model* m, const std::string& name, std::vector<int> shape, double scale) {
    TensorInfo t;
    t.name = name;
    t.ndim = (int)shape.size();
    long n = 1;
    for (int i = 0; i < 4; ++i) {
        t.shape[i] = i < t.ndim ? shape[i] : 1;
        n *= t.shape[i];
    }
    t.off = m->P;
    t.scale = scale;
    m->tensors.push_back(t);
    m->P += n;
    return t.off;
}

static int build_net(mrl_model* m, Net& net, const std::string& prefix) {
    const mrl_model_desc& d = m->d;
    const double s2 = sqrt(2.0);   // a2c/utils.py callers pass init_scale=np.sqrt(2) (f64)
    if (d.network == MRL_NET_LSTM || d.network == MRL_NET_CNN_LSTM) {
        // common/models.py:132-210: [nature_cnn ->] utils.lstm(scope='lstm', nh=nlstm, init_scale=1.0)
        if (!lstm_nh_ok(d.nlstm)) return MRL_EUNSUP;
        int nin;
        if (d.network == MRL_NET_CNN_LSTM) {
            mrl_model_desc dc = d;
            dc.network = MRL_NET_NATURE_CNN;
            m->d = dc;
            int rc = build_net(m, net, prefix);
            m->d = d;
            if (rc) return rc;
            nin = net.nlat;
        } else {
            if (d.ob_dtype != MRL_OB_F32 || m->ob_elems % 4 != 0) return MRL_EUNSUP;    // tf.layers.flatten(X)
            nin = (int)m->ob_elems;
        }
        net.lstm = true; net.lstm_nin = nin; net.nh = d.nlstm;
        if (d.layer_norm) {
            // models.py:173-174 / 200-201: utils.lnlstm(scope='lnlstm'); variables in creation order (a2c/utils.py:113-124):
            // wx, gx (ones), bx, wh, gh (ones), bh, b, gc (ones), bc
            if (!lnlstm_nh_ok(d.nlstm)) return MRL_EUNSUP;
            const std::string sc = prefix + "/lnlstm";
            net.lnl = true;
            net.wx_off = add_tensor(m, sc + "/wx", {nin, 4 * d.nlstm}, 1.0);
            net.gx_off = add_tensor(m, sc + "/gx", {4 * d.nlstm}, -2.0);
            add_tensor(m, sc + "/bx", {4 * d.nlstm}, -1.0);
            net.wh_off = add_tensor(m, sc + "/wh", {d.nlstm, 4 * d.nlstm}, 1.0);
            net.gh_off = add_tensor(m, sc + "/gh", {4 * d.nlstm}, -2.0);
            add_tensor(m, sc + "/bh", {4 * d.nlstm}, -1.0);
            net.lb_off = add_tensor(m, sc + "/b", {4 * d.nlstm}, -1.0);
            net.gc_off = add_tensor(m, sc + "/gc", {d.nlstm}, -2.0);
            add_tensor(m, sc + "/bc", {d.nlstm}, -1.0);
            net.nlat = d.nlstm; net.lat_act = ACT_NONE;
            return 0;
        }
        net.wx_off = add_tensor(m, prefix + "/lstm/wx", {nin, 4 * d.nlstm}, 1.0);
        net.wh_off = add_tensor(m, prefix + "/lstm/wh", {d.nlstm, 4 * d.nlstm}, 1.0);
        net.lb_off = add_tensor(m, prefix + "/lstm/b", {4 * d.nlstm}, -1.0);
        net.nlat = d.nlstm; net.lat_act = ACT_NONE;
        return 0;
    }
    if (d.network == MRL_NET_NATURE_CNN) {
        // any image dtype / channel count (models.py:19 casts and scales whatever comes); uint8 with 4 channels (Atari frame
        // stacks) takes the image-resident first-layer kernels, everything else the generic tiled engine
        if (d.ob_ndim != 3 || d.ob_shape[2] < 1) return MRL_EUNSUP;
        int H = d.ob_shape[0], W = d.ob_shape[1], C = d.ob_shape[2];
        // nature_cnn's stack unless the descriptor carries one (cnn_small, models.py:117-129)
        int nf[4] = {32, 64, 64, 0}, rf[4] = {8, 4, 3, 0}, st[4] = {4, 2, 1, 0}, nconv = 3, fch = 512;
        if (d.nconv != 0) {
            if (d.nconv < 1 || d.nconv > 4 || d.fc_hidden < 1) return MRL_EUNSUP;
            nconv = d.nconv; fch = d.fc_hidden;
            for (int i = 0; i < nconv; ++i) {
                nf[i] = d.convs[i][0]; rf[i] = d.convs[i][1]; st[i] = d.convs[i][2];
                if (nf[i] < 4 || nf[i] % 4 != 0 || rf[i] < 1 || st[i] < 1) return MRL_EUNSUP;
            }
        }
        const bool same = d.conv_pad == 1;
        if (d.conv_pad != 0 && d.conv_pad != 1) return MRL_EUNSUP;
        for (int i = 0; i < nconv; ++i) {
            Layer l{};
            char nm[8];
            snprintf(nm, sizeof nm, "c%d", i + 1);
            l.kind = 0; l.H = H; l.W = W; l.C = C; l.rf = rf[i]; l.stride = st[i]; l.NF = nf[i];
            if (same) {
                // tf.nn.conv2d 'SAME': out = ceil(in / stride); total padding = max((out - 1)*stride + rf - in, 0), smaller half in front
                l.OH = (H + st[i] - 1) / st[i]; l.OW = (W + st[i] - 1) / st[i];
                l.pad_t = std::max((l.OH - 1) * st[i] + rf[i] - H, 0) / 2;
                l.pad_l = std::max((l.OW - 1) * st[i] + rf[i] - W, 0) / 2;
            } else {
                if (H < rf[i] || W < rf[i]) return MRL_EUNSUP;
                l.OH = (H - rf[i]) / st[i] + 1;
                l.OW = (W - rf[i]) / st[i] + 1;
            }
            l.K = rf[i] * rf[i] * C; l.N = nf[i]; l.act = ACT_RELU;
            l.w_off = add_tensor(m, prefix + "/" + nm + "/w", {rf[i], rf[i], C, nf[i]}, s2);
            l.b_off = add_tensor(m, prefix + "/" + nm + "/b", {1, nf[i], 1, 1}, -1.0);
            l.out_elems = (long)l.OH * l.OW * l.NF;
            snprintf(l.name, sizeof l.name, "%s", nm);
            net.L.push_back(l);
            H = l.OH; W = l.OW; C = l.NF;
        }
        Layer f{};
        f.kind = 1; f.K = H * W * C; f.N = fch; f.act = ACT_RELU;
        f.w_off = add_tensor(m, prefix + "/fc1/w", {f.K, f.N}, s2);
        f.b_off = add_tensor(m, prefix + "/fc1/b", {f.N}, -1.0);
        f.out_elems = f.N;
        snprintf(f.name, sizeof f.name, "fc1");
        net.L.push_back(f);
        net.nlat = fch; net.lat_act = ACT_RELU;
        return 0;
    } else if (d.network == MRL_NET_MLP) {
        if (d.ob_dtype != MRL_OB_F32 || d.num_layers < 1 || d.num_layers > 8 || d.num_hidden < 1) return MRL_EUNSUP;
        int nin = (int)m->ob_elems;
        for (int i = 0; i < d.num_layers; ++i) {
            Layer f{};
            f.kind = 1; f.K = nin; f.N = d.num_hidden; f.act = d.activation;
            char nm[32];
            snprintf(nm, sizeof nm, "/mlp_fc%d", i);
            f.w_off = add_tensor(m, prefix + nm + "/w", {f.K, f.N}, s2);
            f.b_off = add_tensor(m, prefix + nm + "/b", {f.N}, -1.0);
            if (d.layer_norm) {
                // models.py:97-98: tf.contrib.layers.layer_norm(h, center=True, scale=True) in the enclosing scope -> variables
                // LayerNorm/beta, LayerNorm/gamma (LayerNorm_1/..., ... for the following layers), created beta first
                char ln[40];
                if (i == 0) snprintf(ln, sizeof ln, "/LayerNorm");
                else snprintf(ln, sizeof ln, "/LayerNorm_%d", i);
                f.ln = true;
                f.beta_off = add_tensor(m, prefix + ln + "/beta", {f.N}, -1.0);      // zeros
                f.gamma_off = add_tensor(m, prefix + ln + "/gamma", {f.N}, -2.0);    // ones
            }
            f.out_elems = f.N;
            snprintf(f.name, sizeof f.name, "mlp_fc%d", i);
            net.L.push_back(f);
            nin = f.N;
        }
        net.nlat = nin; net.lat_act = d.activation;
        return 0;
    }
    return MRL_EUNSUP;
}

extern "C" int mrl_model_create(const mrl_model_desc* desc, mrl_model** out) {
    if (!desc || !out) return MRL_EINVAL;
    if (desc->nact < 1 || desc->ob_ndim < 1 || desc->ob_ndim > 3) return MRL_EINVAL;
    if (desc->pd_kind < MRL_PD_CATEGORICAL || desc->pd_kind > MRL_PD_BERNOULLI) return MRL_EUNSUP;
    if (desc->pd_kind == MRL_PD_MULTICATEGORICAL) {
        if (desc->nsub < 1 || desc->nsub > 16) return MRL_EUNSUP;
        int tot = 0;
        for (int i = 0; i < desc->nsub; ++i) { if (desc->nvec[i] < 1) return MRL_EINVAL; tot += desc->nvec[i]; }
        if (tot != desc->nact) return MRL_EINVAL;
    }
    mrl_model* m = new mrl_model();
    m->d = *desc;
    m->P = 0;
    m->ob_elems = 1;
    for (int i = 0; i < desc->ob_ndim; ++i) m->ob_elems *= desc->ob_shape[i];
    m->vf_copy = desc->value_copy != 0;
    // policies.py:162-165: "recurrent architectures are not supported with value_network=copy yet"
    if (m->vf_copy && (desc->network == MRL_NET_LSTM || desc->network == MRL_NET_CNN_LSTM)) { delete m; return MRL_EUNSUP; }
    int rc = build_net(m, m->pi, "ppo2_model/pi");
    if (rc == 0 && m->vf_copy) rc = build_net(m, m->vf, "ppo2_model/vf");
    if (rc) { delete m; return rc; }
    const int nlat = m->pi.nlat, nlatv = m->vf_copy ? m->vf.nlat : nlat;
    m->head_off = m->P;
    m->has_pi_head = (nlat != desc->nact);   // distributions.py:351-355
    m->pi_w = m->pi_b = m->logstd = -1;
    if (m->has_pi_head) {
        m->pi_w = add_tensor(m, "ppo2_model/pi/w", {nlat, desc->nact}, 0.01);
        m->pi_b = add_tensor(m, "ppo2_model/pi/b", {desc->nact}, -1.0);
    }
    if (desc->pd_kind == MRL_PD_DIAG_GAUSSIAN) m->logstd = add_tensor(m, "ppo2_model/pi/logstd", {1, desc->nact}, -1.0);
    m->vf_w = add_tensor(m, "ppo2_model/vf/w", {nlatv, 1}, 1.0);
    m->vf_b = add_tensor(m, "ppo2_model/vf/b", {1}, -1.0);
    m->HP = (int)(m->P - m->head_off);
    *out = m;
    return 0;
}

extern "C" void mrl_model_destroy(mrl_model* m) { delete m; }
extern "C" int mrl_model_state_size(const mrl_model* m) { return (m && m->pi.lstm) ? 2 * m->pi.nh : 0; }
extern "C" long mrl_model_num_params(const mrl_model* m) { return m ? m->P : 0; }
extern "C" int mrl_model_num_tensors(const mrl_model* m) { return m ? (int)m->tensors.size() : 0; }

extern "C" int mrl_model_tensor_info(const mrl_model* m, int i, char* name, int name_cap, int* ndim, int shape[4],
                                     long* offset, double* init_scale) {
    if (!m || i < 0 || i >= (int)m->tensors.size()) return MRL_EINVAL;
    const TensorInfo& t = m->tensors[i];
    if (name && name_cap > 0) {
        strncpy(name, t.name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (ndim) *ndim = t.ndim;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = t.shape[k];
    if (offset) *offset = t.off;
    if (init_scale) *init_scale = t.scale;
    return 0;
}

// ---- workspace carving ------------------------------------------------------------------
struct NetWs {
    std::vector<float*> h, dz;
    uint16_t* planes;    // bf16 x 3 image of one weight matrix (gemmx6.hip.h), shared by both nets
    long long* dbg;      // timing-experiment stamps (behind the zero page)
    // recurrent cell (lstm.hip.h): x@wx, stored gates / masked state / tanh(c), cell output = policy latent, gradients
    float *zx = nullptr, *gates = nullptr, *cm = nullptr, *hm = nullptr, *tc = nullptr, *hout = nullptr, *dhout = nullptr,
          *dzg = nullptr;
    // layer-normalised cell: normalised values and 1/sqrt(var + e) of LN(x@wx), LN(h@wh), LN(c); gradients w.r.t. the raw
    // h@wh and w.r.t. LN(c)
    float *xhx = nullptr, *isx = nullptr, *xhh = nullptr, *ish = nullptr, *xhc = nullptr, *isc = nullptr, *dzh = nullptr,
          *dcn = nullptr;
    // ReLU bit masks of the conv outputs (1 bit per element, written by the forward epilogues that can, consumed by the
    // position-major data-gradient engine instead of the fp32 activations): mbits[l] may be nullptr; mvalid[l] is set
    // by the forward pass of THIS call when the engine that ran layer l wrote them
    std::vector<uint32_t*> mbits;
    std::vector<char> mvalid;
    // pre-split plane tensors (planes.hip.h) of the layer outputs h[l] and of the pre-activation gradients dz[l]: written by
    // the epilogue of the producing engine, staged by the consumer without split arithmetic.  hp[l] / dzp[l] may be nullptr;
    // hpvalid[l] / dzpvalid[l] are set by THIS call's producer (plane stride = rows * width of that call's batch)
    std::vector<uint16_t*> hp, dzp;
    std::vector<char> hpvalid, dzpvalid;
    // layer normalisation: normalised pre-activations xhat[l] [chunk][N] and 1 / sqrt(var + eps) per row (nullptr: no LN)
    std::vector<float*> xhat, istd;
    float* lat() const { return hout ? hout : h.back(); }
    float* dlat() const { return dhout ? dhout : dz.back(); }
};
struct Ws {
    NetWs pi, vf;
    float* pdparam;      // [chunk][nact]  (act side)
    float* part;         // split-K / bias / head partial slabs
    size_t part_floats;
    double* dscratch;    // adv partials [ADV_G][2] | head stat partials [HEAD_MAXBLK][5] | stats acc [5]
    double* sqpart;      // [SQ_MAX_PART] sum-of-squares partials of the gradient (StepCtx)
    float* advstat;      // [2]
    int32_t* srow;       // [chunk] storage rows of the minibatch samples (env-major index translated)
    float* zeros;        // 256 zero floats (out-of-map taps of the weights-resident data-gradient)
    size_t total;
};

constexpr int ADV_G = 256;
constexpr int SQ_MAX_PART = 8192;            // sum-of-squares partial slots per step
constexpr int SQ_BLOCKS_PER_LAUNCH = 1024;   // reduce_slabs blocks (= slots) per launch while they are collected
constexpr int HEAD_MAXBLK = 512;
constexpr int LN_MAXBLK = 1024;             // layer-norm backward: partial (dbeta | dgamma) slabs
constexpr int SPART_MAX = 2048;             // stat partials: max(HEAD_MAXBLK, MLP_MAX_TILES)
constexpr int WGRAD_TARGET_WGS = 1536;
constexpr int IMGRES_MAX_BLOCKS = 256;      // one persistent workgroup (one partial slab) per CU
constexpr int MLP_MAX_TILES = 2048;         // fused MLP step: one 32-sample tile (one partial slab) per workgroup

// ---- tile variants of the GEMM template (gemm.hip.h) and the process-wide tuning table ----------
enum { V_128x32 = 0, V_256x32, V_128x64_W41, V_128x64_W22, V_256x64, V_128x128, V_COUNT };
enum { V_WRES16 = 100, V_WRES8 = 101 };     // weights-resident engine (wres.hip.h), 16 / 8 waves per CU
enum { V_IMGRES = 102 };                    // image-resident weight-gradient engine (imgres.hip.h)
enum { V_LDSDGRAD = 103 };                  // LDS-resident data-gradient engine (ldsdgrad.hip.h)
static const int kVariantBM[V_COUNT] = {128, 256, 128, 128, 256, 128};
static const int kVariantBN[V_COUNT] = {32, 32, 64, 64, 64, 128};

static std::map<std::string, int>& tune_table() { static std::map<std::string, int> t; return t; }

// ---- process-wide options (engine selection knobs; defaults from the environment on first use) --------
static std::map<std::string, int>& option_table() { static std::map<std::string, int> t; return t; }
static int get_option(const char* name, const char* env, int dflt) {
    auto& t = option_table();
    auto it = t.find(name);
    if (it != t.end()) return it->second;
    const char* ev = getenv(env);
    int v = ev ? atoi(ev) : dflt;
    t[name] = v;
    return v;
}
// fp32 x fp32 GEMM sites that run on the bf16 pipe: 0 = fp32 MFMA (bitwise fmaf chain), 1 = six bf16 products per
// multiply (dropped part < 2^-21 of a product), 2 = eight products (dropped part < 2^-29: below one fp32 rounding).
// Default 2: every product of the update is then at least as accurate as an IEEE fp32 multiply (the sums: DESIGN.md 3.1).
// 0: fp32 MFMA engines; anything else: the split engines (kSplitProducts exact bf16 products per multiply, wres.hip.h).
// (Callers compare with 2, the value the option has carried since the eight-product form became the only tuned one.)
static int f32_split_mode() { return get_option("f32_bf16x6", "MRL_F32_BF16X6", 2) ? 2 : 0; }
// transposed-accumulator epilogues and pre-split activation planes (planes.hip.h), eight-product mode only.  Bits:
//   4 / 8 / 64 (default 76): transposed epilogues (16-byte stores, in-lane ReLU mask words) of the conv forward layers / the
//                            data gradients / the hidden fc layer's forward -- c2.fwd 6.3 -> 5.5 ms, c2.dgrad 6.2 -> 5.75 ms;
//   16: the same for the first conv layer's image-resident kernel (slower: 3.0 -> 3.4 ms, its epilogue is issue-bound);
//   1 / 2 / 32: EXPERIMENT, measured slower -- producers also write plane tensors of h / of dz / of the last layer's dz
//               only, consumers stage them without split arithmetic (1.5x the operand bytes in 64-byte pieces, 2.5x the
//               producers' write traffic: c2.fwd 6.3 -> 9.0 ms).  The plane buffers exist only if the bit is set when the
//               workspace is sized.
static int act_planes_mode() {
    if (f32_split_mode() != 2) return 0;
    if (kExp) return get_option("act_planes", "MRL_ACT_PLANES", 76);
    return get_option("tr_epilogue", "MRL_TR_EPILOGUE", 1) ? 76 : 0;       // product: transposed-accumulator epilogues on / off
}
namespace mrl {
hipError_t raise_lds_limit(const void* kern, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kern})) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({dev, kern});
    return e;
}
}  // namespace mrl

// wave-specialised (producer / consumer) form of the tiled split engines (gemmx6s.hip.h): measured NOT faster than the plain
// form (two waves of one SIMD share its VALU issue and its matrix pipe: profiles/README.md), kept as an experiment knob
// transposed-epilogue launches of the tiled split engine: split-at-the-fragment kernel (gemmx6r.hip.h, option x6_frag) or the
// staged-planes kernel (gemmx6.hip.h); bit-identical results
// does a transposed-epilogue launch of this width take the staged-planes kernel (which can read k-tile-major weight planes)?
static bool x6_tr_staged(int N, const long long* dbg) { return !(x6_frag() && !dbg && N % 32 == 0 && (N <= 64 || (x6_frag() & 8))); }
// mrl_set_option "x6_ktm" [MRL_X6_KTM, 1]: fc weight planes of the tiled split engine in k-tile-major order
static int x6_ktm() { return get_option("x6_ktm", "MRL_X6_KTM", 1); }
template <class AF, class EF>
static hipError_t launch_x6_tr(const AF& af, const uint16_t* Bp, const EF& ef, int M, int N, int K, hipStream_t st, long long* dbg, bool ktm = false,
                               int nz = 1, long zslab = 0) {
    // (N <= 64: the 32 x 128 wave tiles of the fc layer measured slower -- fc1.fwd 1.94 -> 2.17 ms, fc1.dgrad 2.39 -> 2.52 -- unless bit 3 asks)
    if (!x6_tr_staged(N, dbg)) return (ktm || nz > 1) ? hipErrorInvalidValue : launch_gemm_x6r(af, Bp, ef, M, N, K, st);
    return launch_gemm_x6_planes<false, true>(af, 0, Bp, ef, M, N, K, st, dbg, ktm, nz, zslab);
}
static int x6_specialised() { return kExp ? get_option("x6_spec", "MRL_X6_SPEC", 0) : 0; }
// phase-stamp / phase-omission knobs exist in experiment builds only
static int dbg_option(const char* name, const char* env) { return kExp ? get_option(name, env, 0) : 0; }
struct OptionDef { const char* name; const char* env; int dflt; };
static const OptionDef kOptions[] = {
    // ---- product options (include/mrl.h)
    {"u8_bf16x3", "MRL_U8_BF16X3", 1}, {"f32_bf16x6", "MRL_F32_BF16X6", 2}, {"mlp_fused", "MRL_MLP_FUSED", 1},
    {"heads_wave", "MRL_HEADS_WAVE", 2}, {"dgrad_async", "MRL_DGRAD_ASYNC", 1}, {"dgrad_x6", "MRL_DGRAD_X6", 2},
    {"fused_norm", "MRL_FUSED_NORM", 1}, {"relu_bits", "MRL_RELU_BITS", 1}, {"c1_lds", "MRL_C1_LDS", 4},
    {"wgrad_x8", "MRL_WGRAD_X8", 1}, {"c1_wgrad2", "MRL_C1_WGRAD2", 3}, {"wgrad_tr", "MRL_WGRAD_TR", 1},
    {"x6_pg", "MRL_X6_PG", 8}, {"tr_epilogue", "MRL_TR_EPILOGUE", 1}, {"mlp_waves", "MRL_MLP_WAVES", 8},
    {"mlp_slice", "MRL_MLP_SLICE", 1}, {"lstm_e1", "MRL_LSTM_E1", 1}, {"x6_dither", "MRL_X6_DITHER", 3},
    {"x6_frag", "MRL_X6_FRAG", 1}, {"conv_x6c", "MRL_CONV_X6C", 1}, {"wgrad_pipe", "MRL_WGRAD_PIPE", 1}, {"x6_ktm", "MRL_X6_KTM", 1}, {"wgrad_xcd", "MRL_WGRAD_XCD", 1}, {"mlp_act", "MRL_MLP_ACT", 1}, {"x6_splitk", "MRL_X6_SPLITK", 1}, {"dqn_overlap", "MRL_DQN_OVERLAP", 1}, {"gae_lane", "MRL_GAE_LANE", 1}, {"conv_splitk", "MRL_CONV_SPLITK", 1}, {"dqn_latdgrad", "MRL_DQN_LATDGRAD", 1}, {"dqn_wstream", "MRL_DQN_WSTREAM", 1}, {"dqn_heads", "MRL_DQN_HEADS", 1}, {"conv_skinny", "MRL_CONV_SKINNY", 1}, {"dqn_pair", "MRL_DQN_PAIR", 1},
#ifdef MRL_X6_EXPERIMENTS
    // ---- experiment builds only (-DMRL_X6_EXPERIMENTS): measured-and-dropped variants, phase stamps / omissions
    {"imgres_nacc", "MRL_IMGRES_NACC", 0}, {"mlp_dbg", "MRL_MLP_DBG", 0}, {"dgrad_dbg", "MRL_DGRAD_DBG", 0},
    {"x6_dbg", "MRL_X6_DBG", 0}, {"dgx6_dbg", "MRL_DGX6_DBG", 0}, {"x6_spec", "MRL_X6_SPEC", 0}, {"x6_prio", "MRL_X6_PRIO", 0},
    {"c1_dbg", "MRL_C1_DBG", 0}, {"act_planes", "MRL_ACT_PLANES", 76}, {"x6_il", "MRL_X6_IL", 1},
#endif
};
extern "C" int mrl_get_option(const char* name, int* value_out) {
    if (!name || !value_out) return MRL_EINVAL;
    if (!strcmp(name, "f32_products")) { *value_out = kSplitProducts; return 0; }      // read-only: fixed by the build
    for (const OptionDef& o : kOptions)
        if (!strcmp(o.name, name)) { *value_out = get_option(name, o.env, o.dflt); return 0; }
    return MRL_EINVAL;
}
extern "C" int mrl_set_option(const char* name, int value) {
    if (!name) return MRL_EINVAL;
    for (const OptionDef& o : kOptions)
        if (!strcmp(o.name, name)) {
            option_table()[name] = value;
            if (!strcmp(name, "x6_prio")) x6_prio() = value;
            if (!strcmp(name, "x6_dither")) x6_dither() = value;
            if (!strcmp(name, "x6_il")) x6_il() = value;
            if (!strcmp(name, "x6_pg")) x6_pg() = value;
            if (!strcmp(name, "x6_frag")) x6_frag() = value;
            if (!strcmp(name, "conv_x6c")) conv_x6c() = value;
            if (!strcmp(name, "wgrad_pipe")) wgrad_tr_pipe() = value;
            if (!strcmp(name, "wgrad_xcd")) wgrad_tr_xcd() = value;
            if (!strcmp(name, "gae_lane")) gae_lane_form() = value;
            if (!strcmp(name, "lstm_e1")) lstm_e1() = value;
            if (!strcmp(name, "x6_dbg")) x6_xd() = value >= 100 ? value - 100 : 0;
            return 0;
        }
    return MRL_EINVAL;
}
extern "C" int mrl_tune_set(const char* label, int variant) {
    if (!label || (variant >= 2 * V_COUNT && variant != V_WRES16 && variant != V_WRES8 && variant != V_IMGRES && variant != V_LDSDGRAD)) return MRL_EINVAL;
    if (variant < 0) tune_table().erase(label);
    else tune_table()[label] = variant;
    return 0;
}
// default tile choice by GEMM shape; `label` ("c1.fwd", "fc1.wgrad", ...) may override it
static int pick_variant(const char* lname, const char* pass, int M, int N, bool wres_ok = false) {
    if (!tune_table().empty()) {
        auto it = tune_table().find(std::string(lname) + "." + pass);
        if (it != tune_table().end()) return it->second;
    }
    if (wres_ok) return V_WRES16;
    if (N <= 32) return V_128x32;
    if (N <= 64) return V_128x64_W41;
    return V_128x128;
}

struct Split { int nsplit, ksplit; };
static Split pick_split(int variant, int Mp, int Np, long Kp) {
    if (variant >= V_WRES16) variant = V_128x32;
    const int bm = kVariantBM[variant % V_COUNT], bn = kVariantBN[variant % V_COUNT];
    long tiles = (long)((Mp + bm - 1) / bm) * ((Np + bn - 1) / bn);
    long ns = std::max<long>(1, WGRAD_TARGET_WGS / std::max<long>(1, tiles));
    // at least 256 reduction rows per split -- 64 where the whole problem is a few hundred workgroups (latency-bound: more, shorter
    // workgroups per CU cover each other's load round trips; the Q-network's conv weight gradients at batch 32: 44 -> 17 us)
    long maxns = std::max<long>(1, Kp / ((Kp <= 32768 && get_option("conv_skinny", "MRL_CONV_SKINNY", 1)) ? 64 : 256));
    ns = std::min(ns, maxns);
    long ks = (Kp + ns - 1) / ns;
    ks = (ks + 31) / 32 * 32;
    Split s;
    s.ksplit = (int)ks;
    s.nsplit = (int)((Kp + ks - 1) / ks);
    return s;
}
static long max_split_floats(int Mp, int Np, long Kp) {     // worst case over the tile variants
    long worst = 0;
    for (int v = 0; v < V_COUNT; ++v) {
        Split s = pick_split(v, Mp, Np, Kp);
        worst = std::max(worst, (long)s.nsplit * ((long)Mp * Np + Np));
    }
    return worst;
}

static long layer_rows(const Layer& l, int B) { return l.kind == 0 ? (long)B * l.OH * l.OW : (long)B; }

static void carve(const mrl_model* m, int chunk, char* base, Ws& ws) {
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
    size_t part_floats = (size_t)HEAD_MAXBLK * m->HP;
    auto do_net = [&](const Net& net, NetWs& nw) {
        nw.h.clear(); nw.dz.clear(); nw.mbits.clear(); nw.mvalid.clear();
        nw.hp.clear(); nw.dzp.clear(); nw.hpvalid.clear(); nw.dzpvalid.clear();
        nw.xhat.clear(); nw.istd.clear();
        for (const Layer& l : net.L) {
            nw.xhat.push_back(l.ln ? (float*)take((size_t)chunk * l.out_elems * 4) : nullptr);
            nw.istd.push_back(l.ln ? (float*)take((size_t)chunk * 4) : nullptr);
            if (l.ln) part_floats = std::max(part_floats, (size_t)LN_MAXBLK * 2 * l.N);
            const size_t li = nw.h.size();
            // planes of h[l]: a ReLU conv output feeding another layer of the split engines; planes of dz[l]: the A operand of
            // layer l's data gradient (l >= 1)
            // (only when the option asks for them at workspace-query time: an experiment knob, off by default -- measured
            // slower than splitting inside the consumers, profiles/README.md)
            const int pm = act_planes_mode();
            const bool hpl = (pm & 1) && l.kind == 0 && l.act == ACT_RELU && l.NF % 32 == 0 && li + 1 < net.L.size();
            const bool dzpl = (pm & (2 | 32)) && li >= 1 && l.act == ACT_RELU && l.out_elems % 32 == 0 && (l.kind == 0 ? l.NF % 32 == 0 : true);
            nw.hp.push_back(hpl ? (uint16_t*)take((size_t)chunk * l.out_elems * 6) : nullptr);
            nw.dzp.push_back(dzpl ? (uint16_t*)take((size_t)chunk * l.out_elems * 6) : nullptr);
            nw.hpvalid.push_back(0); nw.dzpvalid.push_back(0);
            nw.h.push_back((float*)take((size_t)chunk * l.out_elems * 4));
            nw.dz.push_back((float*)take((size_t)chunk * l.out_elems * 4));
            const bool bits = l.kind == 0 && l.act == ACT_RELU && l.NF % 32 == 0;
            nw.mbits.push_back(bits ? (uint32_t*)take((size_t)chunk * l.out_elems / 8) : nullptr);
            nw.mvalid.push_back(0);
            part_floats = std::max(part_floats, (size_t)max_split_floats(l.K, l.N, layer_rows(l, chunk)));
            if (l.kind == 0) part_floats = std::max(part_floats, (size_t)IMGRES_MAX_BLOCKS * ((size_t)l.K * l.N + l.N));
        }
    };
    do_net(m->pi, ws.pi);
    if (m->vf_copy) do_net(m->vf, ws.vf);
    if (m->pi.lstm) {
        const Net& n = m->pi;
        NetWs& nw = ws.pi;
        const size_t g4 = (size_t)chunk * 4 * n.nh * 4, g1 = (size_t)chunk * n.nh * 4;
        nw.zx = (float*)take(g4); nw.gates = (float*)take(g4); nw.dzg = (float*)take(g4);
        nw.cm = (float*)take(g1); nw.hm = (float*)take(g1); nw.tc = (float*)take(g1); nw.hout = (float*)take(g1);
        nw.dhout = (float*)take(g1);
        part_floats = std::max(part_floats, (size_t)max_split_floats(n.lstm_nin, 4 * n.nh, chunk));
        part_floats = std::max(part_floats, (size_t)max_split_floats(n.nh, 4 * n.nh, chunk));
        if (n.lnl) {
            nw.xhx = (float*)take(g4); nw.xhh = (float*)take(g4); nw.dzh = (float*)take(g4);
            nw.xhc = (float*)take(g1); nw.dcn = (float*)take(g1);
            nw.isx = (float*)take((size_t)chunk * 4); nw.ish = (float*)take((size_t)chunk * 4); nw.isc = (float*)take((size_t)chunk * 4);
            part_floats = std::max(part_floats, (size_t)LN_MAXBLK * 8 * n.nh);
        }
    }
    {   // scratch for the split weight planes of the largest hidden layer (first layers read observations: own engines)
        size_t pb = 0;
        for (const Net* net : {&m->pi, &m->vf})
            for (size_t i = 1; i < net->L.size(); ++i) {
                pb = std::max(pb, gemm_x6_plane_bytes(net->L[i].N, net->L[i].K));
                if (net->L[i].kind == 0)       // data-gradient planes: S*S*C columns x taps^2*NF (dgradx6.hip.h)
                    pb = std::max(pb, gemm_x6_plane_bytes((long)net->L[i].stride * net->L[i].stride * net->L[i].C,
                                                          (long)((net->L[i].rf + net->L[i].stride - 1) / net->L[i].stride) *
                                                              ((net->L[i].rf + net->L[i].stride - 1) / net->L[i].stride) * net->L[i].NF));
            }
        ws.pi.planes = ws.vf.planes = pb ? (uint16_t*)take(pb) : nullptr;
    }
    if (m->d.network == MRL_NET_MLP)
        part_floats = std::max(part_floats, (size_t)std::min<long>(MLP_MAX_TILES, ((long)chunk + 31) / 32) * (size_t)m->P);
    ws.pdparam = (float*)take((size_t)chunk * m->d.nact * 4);
    ws.part = (float*)take(part_floats * 4);
    ws.part_floats = part_floats;
    ws.dscratch = (double*)take((size_t)(ADV_G * 2 + SPART_MAX * 5 + 8) * 8);
    ws.sqpart = (double*)take((size_t)SQ_MAX_PART * 8);
    ws.advstat = (float*)take(64);
    ws.srow = (int32_t*)take((size_t)chunk * 4);
    ws.zeros = (float*)take(2048);
    ws.pi.dbg = ws.vf.dbg = base ? reinterpret_cast<long long*>(ws.zeros) + 64 : nullptr;
    ws.total = off;
}

extern "C" size_t mrl_model_workspace_bytes(const mrl_model* m, int chunk) {
    if (!m || chunk <= 0) return 0;
    Ws ws;
    carve(m, chunk, nullptr, ws);
    return ws.total;
}

// ============================================================================================
// small kernels
// ============================================================================================

// per-call state threaded through the layer launches of one gradient computation
struct StepCtx {
    double* sqpart = nullptr;   // sum-of-squares partials of the FINAL gradient, one per reduce_slabs block (nullptr: not collected)
    int sqcap = 0, sqn = 0;     // capacity / slots used so far
    mrl_comm* comm = nullptr;   // data-parallel all-reduce of finished gradient slices (only on the last chunk)
    float rank_weight = 1.f;
    bool final_chunk = true;
    int early_layer = -1;       // pi-net layer after whose weight gradient [early_lo, P) is complete and can travel
    long early_lo = 0;
    // latency-bound batches (the DQN learner step): the weight gradients -- off the dz chain -- go to this stream, forked per layer
    // with ev_fork once dz[i] is there; ws.part then belongs to that stream and the CALLER joins it before the gradient is read
    // (round-robin over up to three streams, each with its own split-K scratch wpart[j] of ws.part_floats floats, so that the weight
    // gradients of consecutive layers overlap too)
    hipStream_t wstream[3] = {nullptr, nullptr, nullptr};
    float* wpart[3] = {nullptr, nullptr, nullptr};
    int nwstream = 0, wrr = 0;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

// out[i] = (accumulate ? out[i] : 0) + sum_z part[z*slab + i]   (fixed order -> deterministic).
// 64 outputs x 4 z-lanes per block: lane q sums z = q, q+4, ... with 4 loads in flight, the four
// lane sums are combined in a fixed order through LDS.  sq != nullptr: the block also leaves the f64 sum of squares of
// the values it wrote in sq[blockIdx.x] -- the global-norm partials of tf.clip_by_global_norm (model.py:105-107) come
// out of the pass that produces the gradient instead of a second pass over it.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ part, long slab, int nz,
                                                           float* __restrict__ out, long n, int accumulate,
                                                           double* __restrict__ sq, const double* __restrict__ spart,
                                                           int nsp, float invB, float* __restrict__ stats_out) {
    __shared__ float sh[4][64];
    __shared__ double shd[4];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    double ssq = 0.0;
    for (long i0 = blockIdx.x * 64L; i0 < n; i0 += (long)gridDim.x * 64L) {
        const long i = i0 + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (i < n) {
            int zz = q;
            for (; zz + 12 < nz; zz += 16) {
                s0 += part[(long)zz * slab + i];
                s1 += part[(long)(zz + 4) * slab + i];
                s2 += part[(long)(zz + 8) * slab + i];
                s3 += part[(long)(zz + 12) * slab + i];
            }
            for (; zz < nz; zz += 4) s0 += part[(long)zz * slab + i];
        }
        sh[q][c] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (q == 0 && i < n) {
            float t = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
            if (accumulate) t = out[i] + t;
            out[i] = t;
            ssq += (double)t * (double)t;
        }
        __syncthreads();
    }
    if (sq) {
        const double r = block_sum_256(ssq, shd);
        if (threadIdx.x == 0) sq[blockIdx.x] = r;
    }
    // rider: the 5 loss statistics of the step, stats_out[j] = (sum_blk spart[blk][j]) * invB (fixed order)
    if (stats_out && blockIdx.x == 0) {          // wave w sums statistic w (then 4): lanes stride the blocks, fixed shuffle tree
        const int lane = threadIdx.x & 63;
        for (int j = threadIdx.x >> 6; j < 5; j += 4) {
            double t = 0.0;
            for (int b = lane; b < nsp; b += 64) t += spart[b * 5 + j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0) stats_out[j] = (float)(t * (double)invB);
        }
    }
}
static int reduce_slabs(const float* part, long slab, int nz, float* out, long n, int accumulate, hipStream_t st,
                        StepCtx* ctx = nullptr, const double* spart = nullptr, int nsp = 0, float invB = 0.f,
                        float* stats_out = nullptr) {
    int blocks = (int)std::min<long>((n + 63) / 64, 8192);
    double* sq = nullptr;
    if (ctx && ctx->sqpart) {
        blocks = std::min(blocks, SQ_BLOCKS_PER_LAUNCH);
        if (ctx->sqn + blocks > ctx->sqcap) return MRL_ENOSPC;
        sq = ctx->sqpart + ctx->sqn;
        ctx->sqn += blocks;
    }
    ProfScope ps("reduce_slabs", 0.0, 4.0 * n * (nz + 1 + (accumulate ? 1 : 0)), st);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, part, slab, nz, out, n, accumulate, sq, spart, nsp,
                       invB, stats_out);
    MRL_LAUNCH_CHECK();
    return 0;
}

// second half of a split-K forward layer: out[m][n] = act(bias[n] + sum_z part[z][m][n]), z in fixed order (deterministic)
__global__ __launch_bounds__(256) void splitk_bias_act_kernel(const float* __restrict__ part, long slab, int nz,
                                                              const float* __restrict__ bias, int act, float* __restrict__ out,
                                                              long n, int N) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = 0;
        for (; z + 3 < nz; z += 4) {
            s0 += part[(long)z * slab + i]; s1 += part[(long)(z + 1) * slab + i];
            s2 += part[(long)(z + 2) * slab + i]; s3 += part[(long)(z + 3) * slab + i];
        }
        for (; z < nz; ++z) s0 += part[(long)z * slab + i];
        out[i] = act_fwd(((s0 + s1) + (s2 + s3)) + bias[i % N], act);
    }
}

// env-major flat minibatch indices (ppo2.py:160-162, runner.py:69-74) -> time-major storage rows
__global__ __launch_bounds__(256) void translate_idx_kernel(const int64_t* __restrict__ idx, int B, int T, int N,
                                                            int32_t* __restrict__ srow) {
    int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) srow[b] = (int32_t)envmajor_to_row(idx[b], T, N);
}

// advantage statistics over the minibatch (model.py:136-139), f64 accumulation
__global__ __launch_bounds__(256) void advstat_part_kernel(const float* __restrict__ ret, const float* __restrict__ val,
                                                           const int64_t* __restrict__ idx, int B, int T, int N,
                                                           double* __restrict__ part, int32_t* __restrict__ srow_out,
                                                           int srow_cap) {
    __shared__ double sh[4];
    double s = 0.0, s2 = 0.0;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        long r = idx ? envmajor_to_row(idx[b], T, N) : b;
        if (idx && srow_out && b < srow_cap) srow_out[b] = (int32_t)r;     // translated once, reused by the loaders
        float a = __fsub_rn(ret[r], val[r]);
        s += (double)a;
        s2 += (double)a * (double)a;
    }
    double t = block_sum_256(s, sh);
    double t2 = block_sum_256(s2, sh);
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = t; part[blockIdx.x * 2 + 1] = t2; }
}
__global__ __launch_bounds__(256) void advstat_final_kernel(const double* __restrict__ part, int G, int B,
                                                            float* __restrict__ advstat, double* __restrict__ stats_acc) {
    __shared__ double sh[4];
    double s = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < G; i += 256) { s += part[i * 2]; s2 += part[i * 2 + 1]; }
    double t = block_sum_256(s, sh);
    double t2 = block_sum_256(s2, sh);
    if (threadIdx.x == 0) {
        double mean = t / B;
        double var = t2 / B - mean * mean;
        if (var < 0) var = 0;
        advstat[0] = (float)mean;
        advstat[1] = (float)sqrt(var);
    }
    if (threadIdx.x < 5) stats_acc[threadIdx.x] = 0.0;
}

// ---- heads ---------------------------------------------------------------------------------
struct HeadArgs {
    const float* lat; const float* vlat; int nlat, nlatv; int shared; int lat_act, vlat_act;
    const float* Wpi; const float* bpi; const float* logstd; const float* Wvf; const float* bvf;
    int has_pi_head, pd_kind, nact, HP, TS;
    int nsub; int nvec[16];       // MRL_PD_MULTICATEGORICAL: slices of the flat logits
    // training inputs
    const void* actions; const float* returns; const float* values; const float* neglogp;
    const int64_t* idx; long row0; int T, N; int Bc;
    const float* advstat; float cliprange, ent_coef, vf_coef, invB;
    float* dz_pi; float* dz_vf; float* hpart; double* spart;
    // act inputs / outputs
    const float* noise; void* actions_out; float* values_out; float* neglogp_out; float* pdparam_out;
};

struct HeadLds {
    float *lat, *vlat, *Wpi, *Wvf, *bpi, *logstd, *hacc, *pi, *dpi, *dls, *v, *dv;
};
__host__ __device__ static inline size_t head_lds_carve(const HeadArgs& a, bool train, float* base, HeadLds& L) {
    size_t o = 0;
    auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += (n + 3) / 4 * 4; return p; };
    L.lat = take((size_t)a.TS * (a.nlat + 1));
    L.vlat = a.shared ? L.lat : take((size_t)a.TS * (a.nlatv + 1));
    L.Wpi = a.has_pi_head ? take((size_t)a.nlat * a.nact) : nullptr;
    L.Wvf = take(a.nlatv);
    L.bpi = take(a.nact);
    L.logstd = take(a.nact);
    L.pi = take((size_t)a.TS * a.nact);
    L.v = take(a.TS);
    if (train) {
        L.hacc = take(a.HP);
        L.dpi = take((size_t)a.TS * a.nact);
        L.dls = take((size_t)a.TS * a.nact);
        L.dv = take(a.TS);
    } else {
        L.hacc = L.dpi = L.dls = L.dv = nullptr;
    }
    return o * sizeof(float);
}

__device__ __forceinline__ void head_load_weights(const HeadArgs& a, const HeadLds& L) {
    const int tid = threadIdx.x;
    if (a.has_pi_head) {
        for (int q = tid; q < a.nlat * a.nact; q += 256) L.Wpi[q] = a.Wpi[q];
        for (int q = tid; q < a.nact; q += 256) L.bpi[q] = a.bpi[q];
    }
    for (int q = tid; q < a.nlatv; q += 256) L.Wvf[q] = a.Wvf[q];
    if (a.pd_kind == MRL_PD_DIAG_GAUSSIAN)
        for (int q = tid; q < a.nact; q += 256) L.logstd[q] = a.logstd[q];
}

// stage the latent rows of a tile and compute pdparam (logits / mean) and the value
__device__ __forceinline__ void head_tile_forward(const HeadArgs& a, const HeadLds& L, int s0, int ns) {
    const int tid = threadIdx.x;
    for (int q = tid; q < ns * a.nlat; q += 256) {
        int s = q / a.nlat, k = q - s * a.nlat;
        L.lat[s * (a.nlat + 1) + k] = a.lat[(long)(s0 + s) * a.nlat + k];
    }
    if (!a.shared)
        for (int q = tid; q < ns * a.nlatv; q += 256) {
            int s = q / a.nlatv, k = q - s * a.nlatv;
            L.vlat[s * (a.nlatv + 1) + k] = a.vlat[(long)(s0 + s) * a.nlatv + k];
        }
    __syncthreads();
    const int nout = a.nact + 1;
    for (int q = tid; q < ns * nout; q += 256) {
        int s = q / nout, j = q - s * nout;
        if (j < a.nact) {
            float acc;
            if (a.has_pi_head) {
                acc = 0.f;
                const float* x = L.lat + s * (a.nlat + 1);
                for (int k = 0; k < a.nlat; ++k) acc = fmaf(x[k], L.Wpi[k * a.nact + j], acc);
                acc += L.bpi[j];
            } else {
                acc = L.lat[s * (a.nlat + 1) + j];
            }
            L.pi[s * a.nact + j] = acc;
        } else {
            float acc = 0.f;
            const float* x = L.vlat + s * (a.nlatv + 1);
            for (int k = 0; k < a.nlatv; ++k) acc = fmaf(x[k], L.Wvf[k], acc);
            L.v[s] = acc + a.bvf[0];
        }
    }
    __syncthreads();
}

#define MRL_HALF_LOG_2PI 0.9189385332046727f      /* f32(0.5*log(2*pi)) */
#define MRL_HALF_LOG_2PIE 1.4189385332046727f     /* f32(0.5*log(2*pi*e)) */

__global__ __launch_bounds__(256) void heads_train_kernel(HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double sred[4];
    HeadLds L;
    head_lds_carve(a, true, smem, L);
    const int tid = threadIdx.x;
    head_load_weights(a, L);
    for (int q = tid; q < a.HP; q += 256) L.hacc[q] = 0.f;
    double st[5] = {0, 0, 0, 0, 0};
    const int ntiles = (a.Bc + a.TS - 1) / a.TS;
    const float mean = a.advstat[0], sd = a.advstat[1] + 1e-8f;
    const float eps = a.cliprange;
    const float ce = a.ent_coef * a.invB;
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int s0 = tile * a.TS, ns = min(a.TS, a.Bc - s0);
        head_tile_forward(a, L, s0, ns);
        if (tid < ns) {
            const int s = tid;
            const int b = s0 + s;
            const long r = a.idx ? envmajor_to_row(a.idx[b], a.T, a.N) : a.row0 + b;
            const float R = a.returns[r], oldv = a.values[r], oldnlp = a.neglogp[r];
            const float adv = ((R - oldv) - mean) / sd;
            float* pi = L.pi + s * a.nact;
            float* dpi = L.dpi + s * a.nact;
            float nlp, H;
            if (a.pd_kind == MRL_PD_CATEGORICAL) {
                const int act = static_cast<const int32_t*>(a.actions)[r];
                float mx = pi[0];
                for (int j = 1; j < a.nact; ++j) mx = fmaxf(mx, pi[j]);
                float z0 = 0.f;
                for (int j = 0; j < a.nact; ++j) z0 += expf(pi[j] - mx);
                const float logz = logf(z0);
                nlp = logz - (pi[act] - mx);
                H = 0.f;
                for (int j = 0; j < a.nact; ++j) {
                    float a0 = pi[j] - mx;
                    H += (expf(a0) / z0) * (logz - a0);
                }
                const float ratio = expf(oldnlp - nlp);
                const float pg1 = -adv * ratio;
                const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
                const float pg2 = -adv * rc;
                float dr = (pg1 >= pg2) ? -adv : ((ratio >= 1.f - eps && ratio <= 1.f + eps) ? -adv : 0.f);
                const float dnlp = dr * (-ratio) * a.invB;
                for (int j = 0; j < a.nact; ++j) {
                    float a0 = pi[j] - mx;
                    float p = expf(a0) / z0;
                    float logp = a0 - logz;
                    dpi[j] = dnlp * (p - (j == act ? 1.f : 0.f)) + ce * p * (logp + H);
                }
                st[0] += (double)fmaxf(pg1, pg2);
                st[3] += 0.5 * (double)((nlp - oldnlp) * (nlp - oldnlp));
                st[4] += (fabsf(ratio - 1.f) > eps) ? 1.0 : 0.0;
            } else if (a.pd_kind == MRL_PD_MULTICATEGORICAL || a.pd_kind == MRL_PD_BERNOULLI) {
                // distributions.py:206-225 / 253-276: neglogp and entropy are SUMS over independent sub-distributions; the
                // gradient of a slice is the categorical one with that slice's own entropy.  Two passes: the sums first (they
                // fix the ratio), then the logit gradients.
                const bool multi = a.pd_kind == MRL_PD_MULTICATEGORICAL;
                const int nsub = multi ? a.nsub : a.nact;
                const int32_t* x = static_cast<const int32_t*>(a.actions) + r * nsub;
                nlp = 0.f; H = 0.f;
                int o = 0;
                for (int q = 0; q < nsub; ++q) {
                    if (multi) {
                        const int nv = a.nvec[q];
                        const float* l = pi + o;
                        float mx = l[0];
                        for (int j = 1; j < nv; ++j) mx = fmaxf(mx, l[j]);
                        float z0 = 0.f;
                        for (int j = 0; j < nv; ++j) z0 += expf(l[j] - mx);
                        const float logz = logf(z0);
                        // an out-of-range component is an all-zero one_hot row (distributions.py:214): that slice's cross
                        // entropy is 0 and its gradient softmax - 0 (below); never an out-of-bounds read
                        if ((unsigned)x[q] < (unsigned)nv) nlp += logz - (l[x[q]] - mx);
                        float Hq = 0.f;
                        for (int j = 0; j < nv; ++j) { const float a0 = l[j] - mx; Hq += (expf(a0) / z0) * (logz - a0); }
                        H += Hq;
                        o += nv;
                    } else {
                        // tf.nn.sigmoid_cross_entropy_with_logits(l, z) = max(l, 0) - l z + log(1 + exp(-|l|))
                        const float l = pi[q], sp = fmaxf(l, 0.f) + log1pf(expf(-fabsf(l)));
                        const float p = 1.f / (1.f + expf(-l));
                        nlp += sp - l * (float)x[q];
                        H += sp - l * p;
                    }
                }
                const float ratio = expf(oldnlp - nlp);
                const float pg1 = -adv * ratio;
                const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
                const float pg2 = -adv * rc;
                float dr = (pg1 >= pg2) ? -adv : ((ratio >= 1.f - eps && ratio <= 1.f + eps) ? -adv : 0.f);
                const float dnlp = dr * (-ratio) * a.invB;
                o = 0;
                for (int q = 0; q < nsub; ++q) {
                    if (multi) {
                        const int nv = a.nvec[q];
                        const float* l = pi + o;
                        float mx = l[0];
                        for (int j = 1; j < nv; ++j) mx = fmaxf(mx, l[j]);
                        float z0 = 0.f;
                        for (int j = 0; j < nv; ++j) z0 += expf(l[j] - mx);
                        const float logz = logf(z0);
                        float Hq = 0.f;
                        for (int j = 0; j < nv; ++j) { const float a0 = l[j] - mx; Hq += (expf(a0) / z0) * (logz - a0); }
                        for (int j = 0; j < nv; ++j) {
                            const float a0 = l[j] - mx, p = expf(a0) / z0, logp = a0 - logz;
                            dpi[o + j] = dnlp * (p - (j == x[q] ? 1.f : 0.f)) + ce * p * (logp + Hq);
                        }
                        o += nv;
                    } else {
                        // d neglogp / dl = sigmoid(l) - x;  d H / dl = -l p (1 - p)  (the binary entropy through p = sigmoid(l))
                        const float l = pi[q], p = 1.f / (1.f + expf(-l));
                        dpi[q] = dnlp * (p - (float)x[q]) + ce * l * p * (1.f - p);
                    }
                }
                st[0] += (double)fmaxf(pg1, pg2);
                st[3] += 0.5 * (double)((nlp - oldnlp) * (nlp - oldnlp));
                st[4] += (fabsf(ratio - 1.f) > eps) ? 1.0 : 0.0;
            } else {
                const float* x = static_cast<const float*>(a.actions) + r * a.nact;
                float* dls = L.dls + s * a.nact;
                float ssum = 0.f, lsum = 0.f;
                H = 0.f;
                for (int k = 0; k < a.nact; ++k) {
                    float ls = L.logstd[k];
                    float u = (x[k] - pi[k]) / expf(ls);
                    ssum += u * u;
                    lsum += ls;
                    H += ls + MRL_HALF_LOG_2PIE;
                }
                nlp = 0.5f * ssum + MRL_HALF_LOG_2PI * (float)a.nact + lsum;
                const float ratio = expf(oldnlp - nlp);
                const float pg1 = -adv * ratio;
                const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
                const float pg2 = -adv * rc;
                float dr = (pg1 >= pg2) ? -adv : ((ratio >= 1.f - eps && ratio <= 1.f + eps) ? -adv : 0.f);
                const float dnlp = dr * (-ratio) * a.invB;
                for (int k = 0; k < a.nact; ++k) {
                    float sdk = expf(L.logstd[k]);
                    float u = (x[k] - pi[k]) / sdk;
                    dpi[k] = dnlp * (-(u / sdk));
                    dls[k] = dnlp * (1.f - u * u) - ce;
                }
                st[0] += (double)fmaxf(pg1, pg2);
                st[3] += 0.5 * (double)((nlp - oldnlp) * (nlp - oldnlp));
                st[4] += (fabsf(ratio - 1.f) > eps) ? 1.0 : 0.0;
            }
            // value loss (model.py:68-75)
            const float v = L.v[s];
            const float dvc = fminf(fmaxf(v - oldv, -eps), eps);
            const float vclip = oldv + dvc;
            const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
            float dl = (l1 >= l2) ? (v - R) : ((v - oldv >= -eps && v - oldv <= eps) ? (vclip - R) : 0.f);
            L.dv[s] = a.vf_coef * a.invB * dl;
            st[1] += 0.5 * (double)fmaxf(l1, l2);
            st[2] += (double)H;
        }
        __syncthreads();
        // dz of the latent layers: (dpi . Wpi^T [+ dv * Wvf]) * act'(lat)
        for (int q = tid; q < ns * a.nlat; q += 256) {
            int s = q / a.nlat, k = q - s * a.nlat;
            float g;
            if (a.has_pi_head) {
                g = 0.f;
                const float* w = L.Wpi + k * a.nact;
                const float* d = L.dpi + s * a.nact;
                for (int j = 0; j < a.nact; ++j) g = fmaf(d[j], w[j], g);
            } else {
                g = L.dpi[s * a.nact + k];
            }
            if (a.shared) g = fmaf(L.dv[s], L.Wvf[k], g);
            float hv = L.lat[s * (a.nlat + 1) + k];
            a.dz_pi[(long)(s0 + s) * a.nlat + k] = g * act_bwd_from_out(hv, a.lat_act);
        }
        if (!a.shared)
            for (int q = tid; q < ns * a.nlatv; q += 256) {
                int s = q / a.nlatv, k = q - s * a.nlatv;
                float hv = L.vlat[s * (a.nlatv + 1) + k];
                a.dz_vf[(long)(s0 + s) * a.nlatv + k] = L.dv[s] * L.Wvf[k] * act_bwd_from_out(hv, a.vlat_act);
            }
        // head parameter gradients accumulated in LDS (each element owned by one thread)
        for (int e = tid; e < a.HP; e += 256) {
            int q = e;
            float g = 0.f;
            bool done = false;
            if (a.has_pi_head) {
                if (q < a.nlat * a.nact) {
                    int k = q / a.nact, j = q - k * a.nact;
                    for (int s = 0; s < ns; ++s) g = fmaf(L.lat[s * (a.nlat + 1) + k], L.dpi[s * a.nact + j], g);
                    done = true;
                } else {
                    q -= a.nlat * a.nact;
                    if (q < a.nact) {
                        for (int s = 0; s < ns; ++s) g += L.dpi[s * a.nact + q];
                        done = true;
                    } else q -= a.nact;
                }
            }
            if (!done && a.pd_kind == MRL_PD_DIAG_GAUSSIAN) {
                if (q < a.nact) {
                    for (int s = 0; s < ns; ++s) g += L.dls[s * a.nact + q];
                    done = true;
                } else q -= a.nact;
            }
            if (!done) {
                if (q < a.nlatv) {
                    for (int s = 0; s < ns; ++s) g = fmaf(L.vlat[s * (a.nlatv + 1) + q], L.dv[s], g);
                } else {
                    for (int s = 0; s < ns; ++s) g += L.dv[s];
                }
            }
            L.hacc[e] += g;
        }
        __syncthreads();
    }
    for (int e = tid; e < a.HP; e += 256) a.hpart[(long)blockIdx.x * a.HP + e] = L.hacc[e];
    for (int j = 0; j < 5; ++j) {
        double t = block_sum_256(st[j], sred);
        if (tid == 0) a.spart[blockIdx.x * 5 + j] = t;
    }
}

// ---- wave-per-sample variant of heads_train_kernel for the NatureCNN head shape -------------------------------------
// (Categorical, <= 8 actions, value head on the SAME latent, nlat = 64*KPL).  The tile kernel above keeps one workgroup
// per CU busy with scalar LDS loops (1.65 ms per 131072 samples, 0.33 TB/s); here a wave owns a sample at a time:
// lane l holds latent elements [KPL*l, KPL*l + KPL) -- a row is one coalesced 2 KB read -- with its slice of Wpi / Wvf
// in registers; the nact+1 dot products are butterfly-reduced across the wave (every lane ends with the same sums),
// every lane evaluates the per-sample loss algebra of model.py:57-91 redundantly, writes its slice of the masked dz row
// and accumulates its slice of the head-weight gradients in registers.  Fixed summation orders: run-to-run identical.
template <int KPL>
__global__ __launch_bounds__(256) void heads_train_wave_kernel(HeadArgs a) {
    constexpr int NA = 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 waves][HP] head-gradient partials
    __shared__ double sst[4][5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nact = a.nact, nlat = a.nlat;
    float W[KPL][NA], Wv[KPL], gW[KPL][NA], gWv[KPL], gb[NA], gbv = 0.f;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
        const int k = lane * KPL + kk;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            W[kk][j] = j < nact ? a.Wpi[k * nact + j] : 0.f;
            gW[kk][j] = 0.f;
        }
        Wv[kk] = a.Wvf[k];
        gWv[kk] = 0.f;
    }
    float bpi[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) { bpi[j] = j < nact ? a.bpi[j] : 0.f; gb[j] = 0.f; }
    const float bv = a.bvf[0];
    double st[5] = {0, 0, 0, 0, 0};
    const float mean = a.advstat[0], sd = a.advstat[1] + 1e-8f;
    const float eps = a.cliprange;
    const float ce = a.ent_coef * a.invB;
    const int nwaves = gridDim.x * 4;
    for (int b = blockIdx.x * 4 + wave; b < a.Bc; b += nwaves) {
        float x[KPL];
        const float4* src = reinterpret_cast<const float4*>(a.lat + (long)b * nlat + lane * KPL);
#pragma unroll
        for (int q = 0; q < KPL / 4; ++q) {
            const float4 v = src[q];
            x[q * 4 + 0] = v.x; x[q * 4 + 1] = v.y; x[q * 4 + 2] = v.z; x[q * 4 + 3] = v.w;
        }
        const long r = a.idx ? envmajor_to_row(a.idx[b], a.T, a.N) : a.row0 + b;
        const float R = a.returns[r], oldv = a.values[r], oldnlp = a.neglogp[r];
        const int act = static_cast<const int32_t*>(a.actions)[r];
        float pi[NA], v = 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            float t = 0.f;
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk) t = fmaf(x[kk], W[kk][j], t);
            pi[j] = t;
        }
#pragma unroll
        for (int kk = 0; kk < KPL; ++kk) v = fmaf(x[kk], Wv[kk], v);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int j = 0; j < NA; ++j)
                if (j < nact) pi[j] += __shfl_xor(pi[j], off);
            v += __shfl_xor(v, off);
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) pi[j] += bpi[j];
        v += bv;
        // ---- per-sample loss algebra, identical in every lane (same formulas as heads_train_kernel)
        const float adv = ((R - oldv) - mean) / sd;
        float mx = pi[0];
#pragma unroll
        for (int j = 1; j < NA; ++j) if (j < nact) mx = fmaxf(mx, pi[j]);
        float z0 = 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j) if (j < nact) z0 += expf(pi[j] - mx);
        const float logz = logf(z0);
        float pact = 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j) if (j == act) pact = pi[j];
        const float nlp = logz - (pact - mx);
        float H = 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j < nact) {
                const float a0 = pi[j] - mx;
                H += (expf(a0) / z0) * (logz - a0);
            }
        const float ratio = expf(oldnlp - nlp);
        const float pg1 = -adv * ratio;
        const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
        const float pg2 = -adv * rc;
        const float dr = (pg1 >= pg2) ? -adv : ((ratio >= 1.f - eps && ratio <= 1.f + eps) ? -adv : 0.f);
        const float dnlp = dr * (-ratio) * a.invB;
        float dpi[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            dpi[j] = 0.f;
            if (j < nact) {
                const float a0 = pi[j] - mx;
                const float p = expf(a0) / z0;
                const float logp = a0 - logz;
                dpi[j] = dnlp * (p - (j == act ? 1.f : 0.f)) + ce * p * (logp + H);
            }
        }
        const float dvc = fminf(fmaxf(v - oldv, -eps), eps);
        const float vclip = oldv + dvc;
        const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
        const float dl = (l1 >= l2) ? (v - R) : ((v - oldv >= -eps && v - oldv <= eps) ? (vclip - R) : 0.f);
        const float dv = a.vf_coef * a.invB * dl;
        st[0] += (double)fmaxf(pg1, pg2);
        st[1] += 0.5 * (double)fmaxf(l1, l2);
        st[2] += (double)H;
        st[3] += 0.5 * (double)((nlp - oldnlp) * (nlp - oldnlp));
        st[4] += (fabsf(ratio - 1.f) > eps) ? 1.0 : 0.0;
        // ---- dz of the latent layer (masked by act') and this lane's slice of the head gradients
        float g[KPL];
#pragma unroll
        for (int kk = 0; kk < KPL; ++kk) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < NA; ++j) t = fmaf(dpi[j], W[kk][j], t);
            t = fmaf(dv, Wv[kk], t);
            g[kk] = t * act_bwd_from_out(x[kk], a.lat_act);
#pragma unroll
            for (int j = 0; j < NA; ++j) gW[kk][j] = fmaf(x[kk], dpi[j], gW[kk][j]);
            gWv[kk] = fmaf(x[kk], dv, gWv[kk]);
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) gb[j] += dpi[j];
        gbv += dv;
        float4* dst = reinterpret_cast<float4*>(a.dz_pi + (long)b * nlat + lane * KPL);
#pragma unroll
        for (int q = 0; q < KPL / 4; ++q) dst[q] = make_float4(g[q * 4], g[q * 4 + 1], g[q * 4 + 2], g[q * 4 + 3]);
    }
    // ---- combine the 4 waves of the block in fixed order -> one partial slab per block (layout of heads_train_kernel)
    float* mine = smem + (long)wave * a.HP;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
        const int k = lane * KPL + kk;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j < nact) mine[k * nact + j] = gW[kk][j];
        mine[nlat * nact + nact + k] = gWv[kk];
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j < nact) mine[nlat * nact + j] = gb[j];
        mine[nlat * nact + nact + nlat] = gbv;
#pragma unroll
        for (int j = 0; j < 5; ++j) sst[wave][j] = st[j];
    }
    __syncthreads();
    for (int e = tid; e < a.HP; e += 256)
        a.hpart[(long)blockIdx.x * a.HP + e] = ((smem[e] + smem[a.HP + e]) + smem[2 * a.HP + e]) + smem[3 * a.HP + e];
    if (tid < 5) a.spart[blockIdx.x * 5 + tid] = ((sst[0][tid] + sst[1][tid]) + sst[2][tid]) + sst[3][tid];
}

// ---- two samples per wave-step (round 5, `heads_wave` = 2): same per-sample arithmetic, same summation orders -- bit-identical to
// heads_train_wave_kernel -- but a wave-step takes the wave's next TWO samples (b and b + nwaves: the same samples, in the same order,
// the one-sample kernel gives this wave).  The first butterfly step leaves sample A's partial sums in lanes 0..31 and sample B's in
// lanes 32..63, so the remaining five steps reduce both samples at once and the per-sample loss algebra -- which every lane of the
// one-sample kernel evaluates redundantly -- is evaluated ONCE for both (lane half = sample).  The per-sample scalars every lane needs
// afterwards (dpi, dv, the five statistics) come back as wave-uniform values (v_readlane of lane 0 / lane 32) and multiply as scalar
// operands.  Two independent dependency chains per wave also hide each other's latencies (the kernel runs two waves per SIMD).
template <int KPL, int NA>
__global__ __launch_bounds__(256) void heads_train_wave2_kernel(HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [4 waves][HP] head-gradient partials
    __shared__ double sst[4][5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool hiB = lane >= 32;                                     // this lane evaluates sample B's algebra
    const int nact = a.nact, nlat = a.nlat;
    float W[KPL][NA], Wv[KPL], gW[KPL][NA], gWv[KPL], gb[NA], gbv = 0.f;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
        const int k = lane * KPL + kk;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            W[kk][j] = j < nact ? a.Wpi[k * nact + j] : 0.f;
            gW[kk][j] = 0.f;
        }
        Wv[kk] = a.Wvf[k];
        gWv[kk] = 0.f;
    }
    float bpi[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) { bpi[j] = j < nact ? a.bpi[j] : 0.f; gb[j] = 0.f; }
    const float bv = a.bvf[0];
    double st[5] = {0, 0, 0, 0, 0};
    const float mean = a.advstat[0], sd = a.advstat[1] + 1e-8f;
    const float eps = a.cliprange;
    const float ce = a.ent_coef * a.invB;
    const int nwaves = gridDim.x * 4;
    auto bcast = [](float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
    for (int b = blockIdx.x * 4 + wave; b < a.Bc; b += 2 * nwaves) {
        const bool hasB = b + nwaves < a.Bc;                         // wave-uniform
        const int b2 = hasB ? b + nwaves : b;
        float xA[KPL], xB[KPL];
        {
            const float4* sA = reinterpret_cast<const float4*>(a.lat + (long)b * nlat + lane * KPL);
            const float4* sB = reinterpret_cast<const float4*>(a.lat + (long)b2 * nlat + lane * KPL);
#pragma unroll
            for (int q = 0; q < KPL / 4; ++q) {
                const float4 u = sA[q], w = sB[q];
                xA[q * 4 + 0] = u.x; xA[q * 4 + 1] = u.y; xA[q * 4 + 2] = u.z; xA[q * 4 + 3] = u.w;
                xB[q * 4 + 0] = w.x; xB[q * 4 + 1] = w.y; xB[q * 4 + 2] = w.z; xB[q * 4 + 3] = w.w;
            }
        }
        const int bm = hiB ? b2 : b;                                 // the sample this lane's algebra belongs to
        const long r = a.idx ? envmajor_to_row(a.idx[bm], a.T, a.N) : a.row0 + bm;
        const float R = a.returns[r], oldv = a.values[r], oldnlp = a.neglogp[r];
        const int act = static_cast<const int32_t*>(a.actions)[r];
        float pi[NA], v;
        {
            float pA[NA], pB[NA], vA = 0.f, vB = 0.f;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                float t = 0.f, u = 0.f;
#pragma unroll
                for (int kk = 0; kk < KPL; ++kk) { t = fmaf(xA[kk], W[kk][j], t); u = fmaf(xB[kk], W[kk][j], u); }
                pA[j] = t; pB[j] = u;
            }
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk) { vA = fmaf(xA[kk], Wv[kk], vA); vB = fmaf(xB[kk], Wv[kk], vB); }
            // butterfly step 32 for both samples: lane l < 32 ends with A_l + A_{l+32}, lane l >= 32 with B_l + B_{l-32} -- the sums the
            // one-sample kernel's first step forms (own + partner, fp addition commutes)
#pragma unroll
            for (int j = 0; j < NA; ++j)
                if (j < nact) pi[j] = (hiB ? pB[j] : pA[j]) + __shfl_xor(hiB ? pA[j] : pB[j], 32);
                else pi[j] = 0.f;
            v = (hiB ? vB : vA) + __shfl_xor(hiB ? vA : vB, 32);
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
#pragma unroll
            for (int j = 0; j < NA; ++j)
                if (j < nact) pi[j] += __shfl_xor(pi[j], off);
            v += __shfl_xor(v, off);
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) pi[j] += bpi[j];
        v += bv;
        // ---- per-sample loss algebra (formulas and evaluation order of heads_train_wave_kernel), once per lane half
        const float adv = ((R - oldv) - mean) / sd;
        float mx = pi[0];
#pragma unroll
        for (int j = 1; j < NA; ++j) if (j < nact) mx = fmaxf(mx, pi[j]);
        float z0 = 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j) if (j < nact) z0 += expf(pi[j] - mx);
        const float logz = logf(z0);
        float pact = 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j) if (j == act) pact = pi[j];
        const float nlp = logz - (pact - mx);
        float H = 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j < nact) {
                const float a0 = pi[j] - mx;
                H += (expf(a0) / z0) * (logz - a0);
            }
        const float ratio = expf(oldnlp - nlp);
        const float pg1 = -adv * ratio;
        const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
        const float pg2 = -adv * rc;
        const float dr = (pg1 >= pg2) ? -adv : ((ratio >= 1.f - eps && ratio <= 1.f + eps) ? -adv : 0.f);
        const float dnlp = dr * (-ratio) * a.invB;
        float dpi[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            dpi[j] = 0.f;
            if (j < nact) {
                const float a0 = pi[j] - mx;
                const float p = expf(a0) / z0;
                const float logp = a0 - logz;
                dpi[j] = dnlp * (p - (j == act ? 1.f : 0.f)) + ce * p * (logp + H);
            }
        }
        const float dvc = fminf(fmaxf(v - oldv, -eps), eps);
        const float vclip = oldv + dvc;
        const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
        const float dl = (l1 >= l2) ? (v - R) : ((v - oldv >= -eps && v - oldv <= eps) ? (vclip - R) : 0.f);
        const float dv = a.vf_coef * a.invB * dl;
        const float s0 = fmaxf(pg1, pg2), s1 = fmaxf(l1, l2), s3 = (nlp - oldnlp) * (nlp - oldnlp);
        const float s4 = (fabsf(ratio - 1.f) > eps) ? 1.f : 0.f;
        // ---- statistics in the one-sample kernel's order: sample A, then sample B
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && !hasB) continue;
            const int src = half * 32;
            st[0] += (double)bcast(s0, src);
            st[1] += 0.5 * (double)bcast(s1, src);
            st[2] += (double)bcast(H, src);
            st[3] += 0.5 * (double)bcast(s3, src);
            st[4] += (double)bcast(s4, src);
        }
        // ---- dz of the latent layer (masked by act') and this lane's slice of the head gradients: sample A, then sample B
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && !hasB) continue;
            const int src = half * 32;
            float d[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) d[j] = bcast(dpi[j], src);
            const float dvs = bcast(dv, src);
            float g[KPL];
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk) {
                const float xk = half ? xB[kk] : xA[kk];
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < NA; ++j) t = fmaf(d[j], W[kk][j], t);
                t = fmaf(dvs, Wv[kk], t);
                g[kk] = t * act_bwd_from_out(xk, a.lat_act);
#pragma unroll
                for (int j = 0; j < NA; ++j) gW[kk][j] = fmaf(xk, d[j], gW[kk][j]);
                gWv[kk] = fmaf(xk, dvs, gWv[kk]);
            }
#pragma unroll
            for (int j = 0; j < NA; ++j) gb[j] += d[j];
            gbv += dvs;
            float4* dst = reinterpret_cast<float4*>(a.dz_pi + (long)(half ? b2 : b) * nlat + lane * KPL);
#pragma unroll
            for (int q = 0; q < KPL / 4; ++q) dst[q] = make_float4(g[q * 4], g[q * 4 + 1], g[q * 4 + 2], g[q * 4 + 3]);
        }
    }
    // ---- combine the 4 waves of the block in fixed order -> one partial slab per block (as heads_train_wave_kernel)
    float* mine = smem + (long)wave * a.HP;
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
        const int k = lane * KPL + kk;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j < nact) mine[k * nact + j] = gW[kk][j];
        mine[nlat * nact + nact + k] = gWv[kk];
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (j < nact) mine[nlat * nact + j] = gb[j];
        mine[nlat * nact + nact + nlat] = gbv;
#pragma unroll
        for (int j = 0; j < 5; ++j) sst[wave][j] = st[j];
    }
    __syncthreads();
    for (int e = tid; e < a.HP; e += 256)
        a.hpart[(long)blockIdx.x * a.HP + e] = ((smem[e] + smem[a.HP + e]) + smem[2 * a.HP + e]) + smem[3 * a.HP + e];
    if (tid < 5) a.spart[blockIdx.x * 5 + tid] = ((sst[0][tid] + sst[1][tid]) + sst[2][tid]) + sst[3][tid];
}

// stats_acc[j] += sum_blk spart[blk][j].  One workgroup: thread t adds the blocks t, t + 256, ... of statistic j in that order, the 256
// partial sums are combined by a fixed tree (run-to-run identical).  (A single thread per statistic walking all blocks took 60 us per
// minibatch step -- 2.4 % of config 3's update, rocprofv3 trace of the N = 512 row.)
__global__ __launch_bounds__(256) void heads_stats_reduce_kernel(const double* __restrict__ spart, int nblk, double* __restrict__ acc) {
    __shared__ double red[5][256];
    const int t = threadIdx.x;
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = t; b < nblk; b += 256)
#pragma unroll
        for (int j = 0; j < 5; ++j) s[j] += spart[b * 5 + j];
#pragma unroll
    for (int j = 0; j < 5; ++j) red[j][t] = s[j];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w)
#pragma unroll
            for (int j = 0; j < 5; ++j) red[j][t] += red[j][t + w];
        __syncthreads();
    }
    if (t < 5) acc[t] += red[t][0];
}
__global__ void stats_finalize_kernel(const double* __restrict__ acc, float invB, float* __restrict__ out) {
    int j = threadIdx.x;
    if (j < 5) out[j] = (float)(acc[j] * (double)invB);
}

// act side: sample + neglogp (policies.py:52-55)
__global__ __launch_bounds__(256) void heads_act_kernel(HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    HeadLds L;
    head_lds_carve(a, false, smem, L);
    const int tid = threadIdx.x;
    head_load_weights(a, L);
    __syncthreads();
    const int ntiles = (a.Bc + a.TS - 1) / a.TS;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int s0 = tile * a.TS, ns = min(a.TS, a.Bc - s0);
        head_tile_forward(a, L, s0, ns);
        if (tid < ns) {
            const int s = tid;
            const long b = a.row0 + s0 + s;
            const float* pi = L.pi + s * a.nact;
            if (a.values_out) a.values_out[b] = L.v[s];
            if (a.pdparam_out)
                for (int j = 0; j < a.nact; ++j) a.pdparam_out[b * a.nact + j] = pi[j];
            if (a.actions_out) {
                const float* nz = a.noise + b * a.nact;
                if (a.pd_kind == MRL_PD_CATEGORICAL) {
                    // argmax(logits - log(-log(u))), first max wins (tf.argmax)
                    int best = 0;
                    float bv = pi[0] - logf(-logf(nz[0]));
                    for (int j = 1; j < a.nact; ++j) {
                        float c = pi[j] - logf(-logf(nz[j]));
                        if (c > bv) { bv = c; best = j; }
                    }
                    float mx = pi[0];
                    for (int j = 1; j < a.nact; ++j) mx = fmaxf(mx, pi[j]);
                    float z0 = 0.f;
                    for (int j = 0; j < a.nact; ++j) z0 += expf(pi[j] - mx);
                    static_cast<int32_t*>(a.actions_out)[b] = best;
                    a.neglogp_out[b] = logf(z0) - (pi[best] - mx);
                } else if (a.pd_kind == MRL_PD_MULTICATEGORICAL) {
                    int32_t* ao = static_cast<int32_t*>(a.actions_out) + b * a.nsub;
                    float nlp = 0.f;
                    int o = 0;
                    for (int q = 0; q < a.nsub; ++q) {                 // one Gumbel-max per slice (distributions.py:223-224)
                        const int nv = a.nvec[q];
                        const float* l = pi + o;
                        int best = 0;
                        float bv = l[0] - logf(-logf(nz[o]));
                        float mx = l[0];
                        for (int j = 1; j < nv; ++j) {
                            const float c = l[j] - logf(-logf(nz[o + j]));
                            if (c > bv) { bv = c; best = j; }
                            mx = fmaxf(mx, l[j]);
                        }
                        float z0 = 0.f;
                        for (int j = 0; j < nv; ++j) z0 += expf(l[j] - mx);
                        ao[q] = best;
                        nlp += logf(z0) - (l[best] - mx);
                        o += nv;
                    }
                    a.neglogp_out[b] = nlp;
                } else if (a.pd_kind == MRL_PD_BERNOULLI) {
                    int32_t* ao = static_cast<int32_t*>(a.actions_out) + b * a.nact;
                    float nlp = 0.f;
                    for (int q = 0; q < a.nact; ++q) {                 // u < sigmoid(l) (distributions.py:271-273)
                        const float l = pi[q], p = 1.f / (1.f + expf(-l));
                        const int xb = nz[q] < p ? 1 : 0;
                        ao[q] = xb;
                        nlp += fmaxf(l, 0.f) + log1pf(expf(-fabsf(l))) - l * (float)xb;
                    }
                    a.neglogp_out[b] = nlp;
                } else {
                    float* ao = static_cast<float*>(a.actions_out) + b * a.nact;
                    float ssum = 0.f, lsum = 0.f;
                    for (int k = 0; k < a.nact; ++k) {
                        float ls = L.logstd[k], sdk = expf(ls);
                        float x = pi[k] + sdk * nz[k];
                        ao[k] = x;
                        float u = (x - pi[k]) / sdk;
                        ssum += u * u;
                        lsum += ls;
                    }
                    a.neglogp_out[b] = 0.5f * ssum + MRL_HALF_LOG_2PI * (float)a.nact + lsum;
                }
            }
        }
        __syncthreads();
    }
}

// ---- wave-per-sample act heads for the NatureCNN head shape (Categorical <= 8 actions, value head on the same 512-wide latent) ----
// The tile kernel above walks 32 samples per workgroup through LDS scalar loops: at the act batch of a 512-env shard that is 16
// workgroups and 57 us per env step (profiles/r05k_rollout_trace_n512.txt), 12 % of the rollout.  Here a wave owns a sample: lane l
// holds latent elements [8 l, 8 l + 8) and its slice of Wpi / Wvf in registers, the nact + 1 dot products are butterfly-reduced
// (same form as heads_train_wave_kernel, so act and train see the same logits for the same parameters), lane 0 samples
// (Gumbel-max, policies.py:52; distributions.py:199-201) and writes action / value / neglogp.
template <int KPL>
__global__ __launch_bounds__(256) void heads_act_wave_kernel(HeadArgs a) {
    constexpr int NA = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nact = a.nact, nlat = a.nlat;
    float W[KPL][NA], Wv[KPL];
#pragma unroll
    for (int kk = 0; kk < KPL; ++kk) {
        const int k = lane * KPL + kk;
#pragma unroll
        for (int j = 0; j < NA; ++j) W[kk][j] = j < nact ? a.Wpi[k * nact + j] : 0.f;
        Wv[kk] = a.Wvf[k];
    }
    const int nwaves = gridDim.x * 4;
    for (int s = blockIdx.x * 4 + wave; s < a.Bc; s += nwaves) {
        float x[KPL];
        const float4* src = reinterpret_cast<const float4*>(a.lat + (long)s * nlat + lane * KPL);
#pragma unroll
        for (int q = 0; q < KPL / 4; ++q) {
            const float4 v4 = src[q];
            x[q * 4 + 0] = v4.x; x[q * 4 + 1] = v4.y; x[q * 4 + 2] = v4.z; x[q * 4 + 3] = v4.w;
        }
        float pi[NA], v = 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            float t = 0.f;
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk) t = fmaf(x[kk], W[kk][j], t);
            pi[j] = t;
        }
#pragma unroll
        for (int kk = 0; kk < KPL; ++kk) v = fmaf(x[kk], Wv[kk], v);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int j = 0; j < NA; ++j)
                if (j < nact) pi[j] += __shfl_xor(pi[j], off);
            v += __shfl_xor(v, off);
        }
        if (lane == 0) {
            const long b = a.row0 + s;
#pragma unroll
            for (int j = 0; j < NA; ++j) if (j < nact) pi[j] += a.bpi[j];
            if (a.values_out) a.values_out[b] = v + a.bvf[0];
            if (a.pdparam_out) {
#pragma unroll
                for (int j = 0; j < NA; ++j) if (j < nact) a.pdparam_out[b * nact + j] = pi[j];
            }
            if (a.actions_out) {
                const float* nz = a.noise + b * nact;
                int best = 0;                              // argmax(logits - log(-log(u))), first max wins (tf.argmax)
                float bv = pi[0] - logf(-logf(nz[0])), mx = pi[0], pbest = pi[0];
#pragma unroll
                for (int j = 1; j < NA; ++j)
                    if (j < nact) {
                        const float c = pi[j] - logf(-logf(nz[j]));
                        if (c > bv) { bv = c; best = j; pbest = pi[j]; }
                        mx = fmaxf(mx, pi[j]);
                    }
                float z0 = 0.f;
#pragma unroll
                for (int j = 0; j < NA; ++j) if (j < nact) z0 += expf(pi[j] - mx);
                static_cast<int32_t*>(a.actions_out)[b] = best;
                a.neglogp_out[b] = logf(z0) - (pbest - mx);
            }
        }
    }
}

// ============================================================================================
// layer normalisation of mlp(layer_norm=True) --- common/models.py:97-98: tf.contrib.layers.layer_norm(h, center=True,
// scale=True): moments over the features of a row (biased variance), y = (z - mean) * rsqrt(var + 1e-12) * gamma + beta,
// then the activation.  One wave per row; xhat and 1/sqrt(var + eps) are kept for the backward pass.
// ============================================================================================
constexpr float LN_EPS = 1e-12f;
__global__ __launch_bounds__(256) void ln_fwd_kernel(float* __restrict__ h /* in: z, out: act(y) */, float* __restrict__ xhat,
                                                     float* __restrict__ istd, const float* __restrict__ beta,
                                                     const float* __restrict__ gamma, int rows, int N, int act,
                                                     float eps = LN_EPS) {
    const int lane = threadIdx.x & 63;
    for (long r = blockIdx.x * 4L + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4L) {
        float* z = h + r * N;
        float s = 0.f;
        for (int c = lane; c < N; c += 64) s += z[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s / (float)N;
        float q = 0.f;
        for (int c = lane; c < N; c += 64) { const float d = z[c] - mean; q += d * d; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
        const float inv = 1.f / sqrtf(q / (float)N + eps);
        if (lane == 0 && istd) istd[r] = inv;
        for (int c = lane; c < N; c += 64) {
            const float xh = (z[c] - mean) * inv;
            if (xhat) xhat[r * N + c] = xh;
            z[c] = act_fwd(xh * gamma[c] + beta[c], act);
        }
    }
}
// dy (gradient w.r.t. the normalised, scaled and shifted value) -> dz (w.r.t. the affine map's output), in place, and this
// block's partial sums of dbeta[c] = sum_r dy[r][c], dgamma[c] = sum_r dy[r][c] * xhat[r][c] (rows in fixed order)
// MODE 0: as described (shift | gain order of tf.contrib's beta, gamma); MODE 1: the partial slab is (dgain | dshift) -- the
// variable order of lnlstm's gx|bx (a2c/utils.py:114-115); MODE 2: (dgain | dshift) sums only, dz is left alone
template <int MODE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(float* __restrict__ dz, const float* __restrict__ xhat,
                                                     const float* __restrict__ istd, const float* __restrict__ gamma, int rows,
                                                     int N, float* __restrict__ part /* [gridDim.x][2N] */) {
    extern __shared__ float ln_s[];                       // [4 waves][2N]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = ln_s + (long)wave * 2 * N;
    const int o_shift = MODE == 0 ? 0 : N, o_gain = MODE == 0 ? N : 0;
    for (int c = lane; c < 2 * N; c += 64) mine[c] = 0.f;
    for (long r = blockIdx.x * 4L + wave; r < rows; r += (long)gridDim.x * 4L) {
        float* g = dz + r * N;
        const float* xh = xhat + r * N;
        float m1 = 0.f, m2 = 0.f, inv = 0.f;
        if (MODE != 2) {
            float s1 = 0.f, s2 = 0.f;
            for (int c = lane; c < N; c += 64) {
                const float gy = g[c] * gamma[c];
                s1 += gy;
                s2 += gy * xh[c];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
            m1 = s1 / (float)N; m2 = s2 / (float)N; inv = istd[r];
        }
        for (int c = lane; c < N; c += 64) {
            const float dy = g[c], x = xh[c];
            mine[o_shift + c] += dy;
            mine[o_gain + c] += dy * x;
            if (MODE != 2) g[c] = inv * ((dy * gamma[c] - m1) - x * m2);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * N; c += 256)
        part[(long)blockIdx.x * 2 * N + c] = ((ln_s[c] + ln_s[2 * N + c]) + ln_s[4 * N + c]) + ln_s[6 * N + c];
}

// ============================================================================================
// layer launches
// ============================================================================================
struct In {               // layer-0 input description
    const void* obs; const int32_t* srow;   // srow: storage row of sample b (nullptr: b itself)
};

template <class AF, class BF, class EF>
static int gemm_dispatch(const char* lname, const char* pass, int variant, const AF& af, const BF& bf, const EF& ef,
                         int M, int N, int K, int zdim, int ksplit, hipStream_t st, double alg_flops = -1.0) {
    hipError_t e;
    char label[40];
    if (prof_enabled()) snprintf(label, sizeof label, "%s.%s", lname, pass);
    // algorithmic work of the launch: 2*M*N*K flops unless the caller states it (the gather-form conv
    // data-gradient also multiplies zero padding at the image border: only the true
    // 2 * out_pixels * K_conv * NF count is reported)
    ProfScope ps(label, alg_flops >= 0 ? alg_flops : 2.0 * M * (double)N * K * ((ksplit >= K) ? zdim : 1), 0.0, st);
    const bool db = variant >= V_COUNT;     // variants V_COUNT.. are the double-buffered forms
    switch (db ? variant - V_COUNT : variant) {
#ifdef MRL_X6_EXPERIMENTS
#define MRL_CASE(V, WM, WN, TM, TN)                                                                          \
        case V:                                                                                              \
            e = db ? launch_gemm<AF, BF, EF, WM, WN, TM, TN, true>(af, bf, ef, M, N, K, zdim, ksplit, st)     \
                   : launch_gemm<AF, BF, EF, WM, WN, TM, TN, false>(af, bf, ef, M, N, K, zdim, ksplit, st);   \
            break;
#else       // product builds: the single-buffered form of every tile shape (the double-buffered forms measured slower)
#define MRL_CASE(V, WM, WN, TM, TN)                                                                          \
        case V:                                                                                              \
            e = launch_gemm<AF, BF, EF, WM, WN, TM, TN, false>(af, bf, ef, M, N, K, zdim, ksplit, st);        \
            break;
#endif
        MRL_CASE(V_128x32, 4, 1, 1, 1)
        MRL_CASE(V_256x32, 4, 1, 2, 1)
        MRL_CASE(V_128x64_W41, 4, 1, 1, 2)
        MRL_CASE(V_128x64_W22, 2, 2, 2, 1)
        MRL_CASE(V_256x64, 4, 1, 2, 2)
        default:
        MRL_CASE(V_128x128, 2, 2, 2, 2)
#undef MRL_CASE
    }
    return (int)e;
}

static inline int is_vec(const void* p, long ld) { return (ld % 4 == 0) && ((uintptr_t)p % 16 == 0); }

static void fill_conv(ConvGeom& g, const Layer& l, const void* p, int npix, const int32_t* srow) {
    g.p = p; g.H = l.H; g.W = l.W; g.C = l.C; g.rf = l.rf; g.stride = l.stride; g.OH = l.OH; g.OW = l.OW;
    g.npix = npix; g.kconv = l.K; g.srow = srow; g.pad_t = l.pad_t; g.pad_l = l.pad_l;
    g.finish();
}
// SAME-padded layers (taps outside the image) run on the generic tiled engine only
static bool layer_padded(const Layer& l) {
    return l.kind == 0 && (l.pad_t || l.pad_l || (l.OH - 1) * l.stride + l.rf > l.H || (l.OW - 1) * l.stride + l.rf > l.W);
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

// can the weights-resident engine run this conv layer's forward / data-gradient?
constexpr int WRES_PF = 8;
static bool wres_fwd_ok(const Layer& l, bool u8, const void* src) {
    if (l.kind != 0 || layer_padded(l) || l.NF > 64 || wres_lds_bytes(1, l.NF, l.K) > 160 * 1024) return false;
    const int rowk = l.rf * l.C;
    if (u8) return (l.stride * l.C) % 16 == 0 && (l.W * l.C) % 16 == 0 && ((long)l.H * l.W * l.C) % 16 == 0 &&
                   rowk == 32 && l.rf % WRES_PF == 0 && (uintptr_t)src % 16 == 0;
    return l.C % 4 == 0 && rowk % (8 * WRES_PF) == 0 && (uintptr_t)src % 16 == 0;
}
static bool wres_dgrad_ok(const Layer& l) {
    if (l.kind != 0 || layer_padded(l) || l.C > 64 || l.NF != 8 * WRES_PF) return false;
    const int taps = (l.rf + l.stride - 1) / l.stride;
    return wres_lds_bytes(l.stride * l.stride, l.C, taps * taps * l.NF) <= 160 * 1024;
}

template <class AL, class BL, class EF>
static int wres_dispatch(const char* lname, const char* pass, int variant, int ncols, const AL& al, const BL& bl,
                         const EF& ef, int zc, int K, long tiles, double flops, hipStream_t st) {
    char label[40];
    if (prof_enabled()) snprintf(label, sizeof label, "%s.%s", lname, pass);
    ProfScope ps(label, flops, 0.0, st);
    hipError_t e;
    const int ncu = num_cus();
    if (ncols <= 32) {
        e = variant == V_WRES8 ? launch_wres<AL, BL, EF, 1, WRES_PF, 8>(al, bl, ef, zc, K, ncols, tiles, ncu, st)
                               : launch_wres<AL, BL, EF, 1, WRES_PF, 16>(al, bl, ef, zc, K, ncols, tiles, ncu, st);
    } else {    // two accumulator tiles + two fragment sets need > 128 VGPRs: 8 waves (2 per SIMD)
        e = launch_wres<AL, BL, EF, 2, WRES_PF, 8>(al, bl, ef, zc, K, ncols, tiles, ncu, st);
    }
    return (int)e;
}

// image-resident weight gradient: compiled for the NatureCNN geometries (imgres.hip.h is templated on
// the layer shape so that all im2col addressing folds into instruction immediates)
static int imgres_kind(const Layer& l, bool u8, const void* src) {
    if (l.kind != 0 || layer_padded(l) || (uintptr_t)src % 16 != 0) return 0;
    if (u8 && l.H == 84 && l.W == 84 && l.C == 4 && l.rf == 8 && l.stride == 4 && l.NF == 32) return 1;
    if (!u8 && l.H == 20 && l.W == 20 && l.C == 32 && l.rf == 4 && l.stride == 2 && l.NF == 64) return 2;
    if (!u8 && l.H == 9 && l.W == 9 && l.C == 64 && l.rf == 3 && l.stride == 1 && l.NF == 64) return 3;
    return 0;
}
static int imgres_dispatch(int kind, const Layer& l, const void* x, const int32_t* srow, const float* dz,
                           const float* hcur, int B, float* part, int nblocks, hipStream_t st) {
    char label[40];
    if (prof_enabled()) snprintf(label, sizeof label, "%s.wgrad", l.name);
    ProfScope ps(label, 2.0 * B * l.OH * l.OW * (double)l.K * l.NF, 0.0, st);
    hipError_t e;
    const int nacc = kExp ? get_option("imgres_nacc", "MRL_IMGRES_NACC", 0) : 0;   // accumulator replicas per wave (experiment knob)
    const int x3 = get_option("u8_bf16x3", "MRL_U8_BF16X3", 1);          // 0: fp32 MFMA path for the u8 layer
    if (kind == 1 && x3 && !hcur && get_option("c1_wgrad2", "MRL_C1_WGRAD2", 3)) {
        e = launch_c1wgrad(x, srow, dz, B, part, nblocks, st, std::max(0, dbg_option("c1_dbg", "MRL_C1_DBG") - 32));       // both operands transposed while staged (c1wgrad.hip.h)
    } else if (kind == 1 && x3 && !hcur) {
        e = launch_imgres_u8x3_wgrad<84, 84, 4, 8, 4, 32>(x, srow, dz, B, part, nblocks, st);
    } else if (kind == 1) {
        if (nacc == 1) e = launch_imgres_wgrad<true, 84, 84, 4, 8, 4, 32, 8, 1, 1, 1>(x, srow, dz, hcur, B, part, nblocks, st);
        else if (nacc == 2) e = launch_imgres_wgrad<true, 84, 84, 4, 8, 4, 32, 8, 1, 1, 2>(x, srow, dz, hcur, B, part, nblocks, st);
        else e = launch_imgres_wgrad<true, 84, 84, 4, 8, 4, 32, 8, 1, 1, 4>(x, srow, dz, hcur, B, part, nblocks, st);
    } else if (kind == 2) {
        if (nacc == 2) e = launch_imgres_wgrad<false, 20, 20, 32, 4, 2, 64, 8, 2, 2, 2>(x, srow, dz, hcur, B, part, nblocks, st);
        else e = launch_imgres_wgrad<false, 20, 20, 32, 4, 2, 64, 8, 2, 2, 1>(x, srow, dz, hcur, B, part, nblocks, st);
    } else {
        if (nacc == 2) e = launch_imgres_wgrad<false, 9, 9, 64, 3, 1, 64, 12, 3, 1, 2>(x, srow, dz, hcur, B, part, nblocks, st);
        else e = launch_imgres_wgrad<false, 9, 9, 64, 3, 1, 64, 12, 3, 1, 1>(x, srow, dz, hcur, B, part, nblocks, st);
    }
    return (int)e;
}

static bool tuned(const Layer& l, const char* pass);
// the skinny-tile conv kernels (convskinny.hip.h) take over from the tiled engine where that one is latency-bound
static bool conv_skinny_ok(const Layer& l, int npix, const void* src, const char* pass) {
    if (!get_option("conv_skinny", "MRL_CONV_SKINNY", 1) || tuned(l, pass)) return false;
    if (l.C % 4 || l.K % 8 || l.NF % 32 || (uintptr_t)src % 16) return false;
    return ((long)npix + 127) / 128 * ((l.NF + 63) / 64) <= 2L * num_cus();
}

template <bool EXP>
static int layer_forward(const mrl_model* m, const Layer& l, bool first, const In& in, const float* hprev,
                         const float* params, float* hout, uint16_t* planes, long long* dbgbuf, int B, hipStream_t st,
                         uint32_t* mbits = nullptr, char* mwrote = nullptr, const uint16_t* hprev_p = nullptr,
                         uint16_t* hp_out = nullptr, char* hpwrote = nullptr, float* part = nullptr, size_t part_floats = 0) {
    // part: scratch for split-K partial products (small-batch fully connected layers), part_floats of it
    // hprev_p: plane tensor of hprev written by the layer below in THIS call (plane stride B * K elements for an fc layer,
    // B * H*W*C for a conv layer); hp_out: where this layer may leave the plane tensor of its own output
    if (mwrote) *mwrote = 0;
    if (hpwrote) *hpwrote = 0;
    if (!EXP || !(act_planes_mode() & 1)) { hprev_p = nullptr; hp_out = nullptr; }
    const bool tr_plain = (act_planes_mode() & 4) != 0;        // transposed-accumulator epilogues without plane output
    if (!get_option("relu_bits", "MRL_RELU_BITS", 1)) mbits = nullptr;
    const float* W = params + l.w_off;
    const float* bias = params + l.b_off;
    RowMC bf{W, l.N, l.N, l.K, is_vec(W, l.N), nullptr};
    if (l.kind == 0) {
        int npix = B * l.OH * l.OW;
        const void* src = first ? in.obs : (const void*)hprev;
        const bool u8in = first && m->d.ob_dtype == MRL_OB_U8, f32in = first && !u8in;     // f32in: float image, scaled by 1/255 in the loader
        const bool wok = !f32in && wres_fwd_ok(l, u8in, src);
        int var = pick_variant(l.name, "fwd", npix, l.NF, wok);
        if (var >= V_WRES16 && wok) {
            const long tiles = ((long)npix + 31) / 32;
            const double fl = 2.0 * npix * (double)l.K * l.NF;
            WresFwdB wb{W, l.NF};
            WresEpiBiasAct we{hout, l.NF, bias, l.act, (long)npix, l.NF};
            if (first) {
                WresFwdA<true> wa;
                fill_conv(wa, l, in.obs, npix, in.srow);
                const int x3 = get_option("u8_bf16x3", "MRL_U8_BF16X3", 1);   // 0: fp32 MFMA path (bitwise fmaf chain)
                if (x3 && l.NF == 32 && l.K % 256 == 0) {
                    char label[40];
                    if (prof_enabled()) snprintf(label, sizeof label, "%s.fwd", l.name);
                    ProfScope ps(label, fl, 0.0, st);
                    // NatureCNN's first layer: image-resident kernel (c1fwd.hip.h) -- images reach LDS once, coalesced,
                    // instead of 16-byte per-lane gathers of overlapping patches
                    if (l.H == C1_H && l.W == C1_W && l.C == C1_C && l.rf == C1_RF && l.stride == C1_S && l.NF == C1_NF &&
                        l.act == ACT_RELU && get_option("c1_lds", "MRL_C1_LDS", 4)) {
                        if (mbits) { if (mwrote) *mwrote = 1; }
                        if ((get_option("c1_lds", "MRL_C1_LDS", 4) & 15) >= 3 && !hp_out && !(act_planes_mode() & 16))
                            return (int)launch_c1fwd3(in.obs, in.srow, W, bias, hout, mbits, B, num_cus(), st, get_option("c1_lds", "MRL_C1_LDS", 4) >> 4,
                                                      (get_option("c1_lds", "MRL_C1_LDS", 4) & 15) >= 4);   // software-pipelined phases (4: transposed accumulators)
                        if ((get_option("c1_lds", "MRL_C1_LDS", 4) & 15) >= 2) {
                            if (hp_out && hpwrote) *hpwrote = 1;
                            return (int)launch_c1fwd2(in.obs, in.srow, W, bias, hout, mbits, B, num_cus(), st, hp_out,
                                                      (long)B * l.out_elems, (act_planes_mode() & 16) != 0);
                        }
                        return (int)launch_c1fwd_lds(in.obs, in.srow, W, bias, hout, mbits, B, num_cus(), st,
                                                     dbg_option("c1_dbg", "MRL_C1_DBG"));
                    }
                    if (mbits && l.NF == 32 && l.act == ACT_RELU) { we.mask = mbits; if (mwrote) *mwrote = 1; }
                    return (int)launch_wres_u8x3<WresEpiBiasAct, WRES_PF, 16>(wa, W, we, l.K, l.NF, tiles, num_cus(), st);
                }
                return wres_dispatch(l.name, "fwd", var, l.NF, wa, wb, we, 1, l.K, tiles, fl, st);
            } else {
                // fp32 activations: bf16 x 6 split engine (wres.hip.h) unless MRL_F32_BF16X6=0 asks for the fmaf chain
                // fp32 activations: tiled bf16 x 6 engine with the im2col row map (gemmx6.hip.h) unless MRL_F32_BF16X6=0 asks
                // for the fp32-MFMA weights-resident path
                const int x6 = f32_split_mode();
                const int rowk = l.rf * l.C;
                if (x6 && planes && rowk % X6_BK == 0 && l.C % 4 == 0 && (uintptr_t)hprev % 16 == 0) {
                    X6ConvA ca;
                    fill_conv(ca, l, hprev, npix, nullptr);
                    char label[40];
                    if (prof_enabled()) snprintf(label, sizeof label, "%s.fwd", l.name);
                    ProfScope ps(label, fl, 0.0, st);
                    const bool pa = EXP && hprev_p && x6 == 2 && l.C % 32 == 0 && !x6_specialised();
                    const bool trp = EXP && hp_out && hpwrote;   // the epilogue also leaves the plane tensor
                    const bool tr = (trp || tr_plain) && x6 == 2 && l.NF % 32 == 0 && l.act == ACT_RELU && !x6_specialised() &&
                                    (uintptr_t)bias % 16 == 0;
                    // k steps in class-major order (gemmx6.hip.h conv_kcls_*): the split-at-the-fragment launches only, so that
                    // x6_frag = 0 stays the round-4 engine for A/B runs
                    const bool kcls = tr && !pa && x6_frag() && !(x6_frag() & 4) && l.rf % l.stride == 0 && l.C % 32 == 0 && l.NF % 32 == 0;
                    // (the class-resident kernel lays out its own fragment-ordered weight planes)
                    const bool x6c = tr && !pa && !trp && conv_x6c() && B >= 96 && l.NF == 64 && l.pad_t == 0 && l.pad_l == 0 &&
                                     ((l.H == 20 && l.W == 20 && l.C == 32 && l.rf == 4 && l.stride == 2) ||
                                      (l.H == 9 && l.W == 9 && l.C == 64 && l.rf == 3 && l.stride == 1));
                    hipError_t e = x6c ? hipSuccess
                                 : kcls ? launch_split_planes(W, l.K, l.NF, true, planes, st, false, l.rf, l.stride, l.C)
                                        : launch_split_planes(W, l.K, l.NF, true, planes, st, pa);
                    if (e != hipSuccess) return (int)e;
                    ca.kcls = kcls ? 1 : 0;
                    if (pa || tr) {
                        // pre-split operands: A staged from the plane tensor of the layer below, and / or the epilogue leaves
                        // the plane tensor of this layer's output for the layer above
                        const long aps = (long)B * l.H * l.W * l.C;
                        if (pa) ca.p = hprev_p;
                        if (tr) {
                            TrBiasRelu tf{hout, l.NF, bias, nullptr, trp ? hp_out : nullptr, (long)npix * l.NF};
                            if (mbits) { tf.mask = mbits; if (mwrote) *mwrote = 1; }
                            if (trp) *hpwrote = 1;
                            // NatureCNN's conv2 / conv3 at minibatch sizes: class-resident kernel (convx6c.hip.h) -- whole images per
                            // tile, every input element loaded once, 4 / 2 barrier pairs per tile instead of 16 / 18
                            if (x6c) {
                                if (l.H == 20 && l.W == 20 && l.C == 32 && l.rf == 4 && l.stride == 2)
                                    return (int)launch_conv_x6c<20, 20, 32, 4, 2, 64, 3>(hprev, W, planes, tf, B, num_cus(), st);
                                if (l.H == 9 && l.W == 9 && l.C == 64 && l.rf == 3 && l.stride == 1)
                                    return (int)launch_conv_x6c<9, 9, 64, 3, 1, 64, 5>(hprev, W, planes, tf, B, num_cus(), st);
                            }
                            // MRL_X6_DBG = 10 + layer index: phase stamps of that conv layer's forward land behind the zero page
                            long long* dbgq = dbg_option("x6_dbg", "MRL_X6_DBG") == 10 + (int)(&l - &m->pi.L[0]) ? dbgbuf : nullptr;
                            if constexpr (EXP) {
                                if (pa) return (int)launch_gemm_x6_planes<true, true>(ca, aps, planes, tf, npix, l.NF, l.K, st, dbgq);
                            }
                            return (int)launch_x6_tr(ca, planes, tf, npix, l.NF, l.K, st, dbgq);
                        }
                        if constexpr (EXP) {
                            EpiBiasAct efp{hout, l.NF, bias, l.act};
                            if (mbits && l.NF % 32 == 0 && l.act == ACT_RELU) { efp.mask = mbits; if (mwrote) *mwrote = 1; }
                            return (int)launch_gemm_x6_planes<true, false>(ca, aps, planes, efp, npix, l.NF, l.K, st);
                        }
                    }
                    EpiBiasAct efx{hout, l.NF, bias, l.act};
                    if (mbits && l.NF % 32 == 0 && l.act == ACT_RELU) { efx.mask = mbits; if (mwrote) *mwrote = 1; }
#ifdef MRL_X6_EXPERIMENTS
                    if (x6_specialised()) {
                        // MRL_X6_DBG = 10 + layer index: phase stamps of that conv layer's forward land behind the zero page
                        long long* dbgp = dbg_option("x6_dbg", "MRL_X6_DBG") == 10 + (int)(&l - &m->pi.L[0]) ? dbgbuf : nullptr;
                        return (int)launch_gemm_x6s(ca, planes, efx, npix, l.NF, l.K, num_cus(), x6 == 2, st, dbgp);
                    }
#endif
                    return (int)launch_gemm_x6(ca, planes, efx, npix, l.NF, l.K, st,
                                               dbg_option("x6_dbg", "MRL_X6_DBG") == 10 + (int)(&l - &m->pi.L[0]) ? dbgbuf : nullptr, x6 == 2);
                }
                WresFwdA<false> wa;
                fill_conv(wa, l, hprev, npix, nullptr);
                return wres_dispatch(l.name, "fwd", var, l.NF, wa, wb, we, 1, l.K, tiles, fl, st);
            }
        }
        if (var >= V_WRES16) var = l.NF <= 32 ? V_128x32 : V_128x64_W41;
        // latency-bound sizes (a few hundred 128-row tiles at most: the Q-network's layers at learner / actor batches): register-direct
        // skinny tiles (convskinny.hip.h)
        if (conv_skinny_ok(l, npix, src, "fwd")) {
            char label[40];
            if (prof_enabled()) snprintf(label, sizeof label, "%s.fwd", l.name);
            ProfScope ps(label, 2.0 * npix * (double)l.K * l.NF, 0.0, st);
            ConvGeom cg;
            fill_conv(cg, l, src, npix, first ? in.srow : nullptr);
            return (int)(f32in ? launch_conv_skinny_fwd<2>(cg, W, bias, hout, l.NF, l.act, st)
                         : first ? launch_conv_skinny_fwd<1>(cg, W, bias, hout, l.NF, l.act, st)
                                 : launch_conv_skinny_fwd<0>(cg, W, bias, hout, l.NF, l.act, st));
        }
        EpiBiasAct ef{hout, l.NF, bias, l.act};
        if (f32in) {
            ConvPatchKC<2> af;
            fill_conv(af, l, in.obs, npix, in.srow);
            return gemm_dispatch(l.name, "fwd", var, af, bf, ef, npix, l.NF, l.K, 1, l.K, st);
        } else if (first) {
            ConvPatchKC<1> af;
            fill_conv(af, l, in.obs, npix, in.srow);
            return gemm_dispatch(l.name, "fwd", var, af, bf, ef, npix, l.NF, l.K, 1, l.K, st);
        } else {
            ConvPatchKC<false> af;
            fill_conv(af, l, hprev, npix, nullptr);
            // small batches (round 6; the Q-network's conv2 / conv3 at batch 32-64: 13-41 workgroups walking K = 512 / 576 alone, 48 us
            // each, a quarter of the learner step): split K over the z dimension into partial slabs, then the bias / activation pass of
            // the small-batch fc path adds them in fixed order
            const int bm = 128, bn = (var == V_128x32) ? 32 : 64;
            const long tiles = (long)((npix + bm - 1) / bm) * ((l.NF + bn - 1) / bn);
            if (part && tiles * 4 <= num_cus() && l.K >= 256 && l.K % 32 == 0 && get_option("conv_splitk", "MRL_CONV_SPLITK", 1)) {
                int ns = (int)std::min<long>(std::min<long>(8, l.K / 128), num_cus() / std::max<long>(1, tiles));
                const int ksplit = ((l.K + ns - 1) / ns + 31) / 32 * 32;
                ns = (l.K + ksplit - 1) / ksplit;
                const long slab = (long)npix * l.NF;
                if (ns > 1 && (size_t)ns * slab <= part_floats) {
                    EpiPartialPlain ep{part, slab, l.NF};
                    int rc = gemm_dispatch(l.name, "fwd", var, af, bf, ep, npix, l.NF, l.K, ns, ksplit, st, 2.0 * npix * (double)l.K * l.NF);
                    if (rc) return rc;
                    ProfScope ps("splitk_bias_act", 0.0, 4.0 * slab * (ns + 1), st);
                    hipLaunchKernelGGL(splitk_bias_act_kernel, dim3((unsigned)std::min<long>((slab + 255) / 256, 1024)), dim3(256), 0, st,
                                       part, slab, ns, bias, l.act, hout, slab, l.NF);
                    return (int)hipGetLastError();
                }
            }
            return gemm_dispatch(l.name, "fwd", var, af, bf, ef, npix, l.NF, l.K, 1, l.K, st);
        }
    } else {
        int var = pick_variant(l.name, "fwd", B, l.N);
        EpiBiasAct ef{hout, l.N, bias, l.act};
        // Small batch, long rows (the Q-network heads at batch 32: 32 x 7744 @ 7744 x 256): an unsplit GEMM is TWO workgroups
        // walking K = 7744 (0.7 ms).  Split K over ~512 workgroups into partial slabs, then one pass adds them in fixed order,
        // adds the bias and applies the activation.
        const RowKC af0{first ? (const float*)in.obs : hprev, l.K, B, l.K, is_vec(first ? in.obs : (const void*)hprev, l.K),
                        first ? in.srow : nullptr};
        // Round 5: also the act batches of an env shard (B = 256 .. 1023 rows of fc1: 16 tiles walking K = 3136 took 286 us of a
        // 483 us env step, profiles/r05k_rollout_trace_n512.txt); the tiled split engine takes over from B = 1024.
        if (part && B < 1024 && l.K >= 1024 && !tuned(l, "fwd")) {
            const int ntile = ((B + 127) / 128) * ((l.N + 31) / 32);
            int ns = std::max(1, std::min(512 / ntile, l.K / 128));
            const int ksplit = ((l.K + ns - 1) / ns + 31) / 32 * 32;
            ns = (l.K + ksplit - 1) / ksplit;
            const long slab = (long)B * l.N;
            if (ns > 1 && (size_t)ns * slab <= part_floats) {
                EpiPartialPlain ep{part, slab, l.N};
                int rc = gemm_dispatch(l.name, "fwd", V_128x32, af0, bf, ep, B, l.N, l.K, ns, ksplit, st, 2.0 * B * (double)l.K * l.N);
                if (rc) return rc;
                ProfScope ps("splitk_bias_act", 0.0, 4.0 * slab * (ns + 1), st);
                hipLaunchKernelGGL(splitk_bias_act_kernel, dim3((unsigned)std::min<long>((slab + 255) / 256, 1024)), dim3(256), 0, st,
                                   part, slab, ns, bias, l.act, hout, slab, l.N);
                return (int)hipGetLastError();
            }
        }
        if (first) {
            RowKC af{(const float*)in.obs, l.K, B, l.K, is_vec(in.obs, l.K), in.srow};
            return gemm_dispatch(l.name, "fwd", var, af, bf, ef, B, l.N, l.K, 1, l.K, st);
        } else {
            // hidden fc layer: bf16 x 6 tiled engine (gemmx6.hip.h) on large batches unless MRL_F32_BF16X6=0
            if (planes && B >= 1024 && l.N >= 128 && gemm_x6_ok(hprev, l.K, l.K) && !tuned(l, "fwd") &&
                f32_split_mode()) {
                char label[40];
                if (prof_enabled()) snprintf(label, sizeof label, "%s.fwd", l.name);
                ProfScope ps(label, 2.0 * B * (double)l.K * l.N, 0.0, st);
                const bool pa = EXP && hprev_p && f32_split_mode() == 2 && l.K % 32 == 0 && !x6_specialised();
                const bool trf = (act_planes_mode() & 64) && f32_split_mode() == 2 && l.act == ACT_RELU && l.N % 32 == 0 && !x6_specialised() &&
                                 (uintptr_t)bias % 16 == 0;
                long long* const dbgq0 = dbg_option("x6_dbg", "MRL_X6_DBG") == 1 ? dbgbuf : nullptr;
                const bool ktm = trf && !pa && x6_ktm() && x6_tr_staged(l.N, dbgq0);
                hipError_t e = launch_split_planes(W, l.K, l.N, true, planes, st, pa, 0, 1, 32, ktm);        // B[n][k] = W[k][n]
                if (e != hipSuccess) return (int)e;
                if constexpr (EXP) {
                    if (pa && !trf)
                        return (int)launch_gemm_x6_planes<true, false>(X6DenseA{reinterpret_cast<const float*>(hprev_p), (long)l.K},
                                                                       (long)B * l.K, planes, ef, B, l.N, l.K, st);
                }
                if (trf) {      // transposed-accumulator epilogue (16-byte stores), no mask / planes needed above an fc layer
                    TrBiasRelu tf{hout, l.N, bias, nullptr, nullptr, 0};
                    long long* dbgq = dbg_option("x6_dbg", "MRL_X6_DBG") == 1 ? dbgbuf : nullptr;
                    // act batches (round 6): 128 x 128 tiles of a 4096-sample step are 128 workgroups for 512 slots, each walking all
                    // of K = 3136 alone (163 us per env step of the N = 4096 rollout).  Split K over blockIdx.y into partial slabs and
                    // finish with the bias / ReLU pass of the small-batch path.
                    {
                        const long tiles = (long)((B + 127) / 128) * ((l.N + 127) / 128);
                        int nz = (int)std::min<long>(4, (2L * num_cus()) / std::max<long>(1, tiles));
                        nz = std::min(nz, l.K / 512);
                        const long slabz = (long)B * l.N;
                        if (nz >= 2 && !pa && !dbgq && part && (size_t)nz * slabz <= part_floats && x6_tr_staged(l.N, dbgq) &&
                            get_option("x6_splitk", "MRL_X6_SPLITK", 1)) {
                            TrPartial tp{part, l.N};
                            hipError_t e2 = launch_x6_tr(X6DenseA{hprev, (long)l.K}, planes, tp, B, l.N, l.K, st, nullptr, ktm, nz, slabz);
                            if (e2 != hipSuccess) return (int)e2;
                            hipLaunchKernelGGL(splitk_bias_act_kernel, dim3((unsigned)std::min<long>((slabz + 255) / 256, 2048)), dim3(256), 0, st,
                                               part, slabz, nz, bias, l.act, hout, slabz, l.N);
                            return (int)hipGetLastError();
                        }
                    }
                    if constexpr (EXP) {
                        if (pa) return (int)launch_gemm_x6_planes<true, true>(X6DenseA{reinterpret_cast<const float*>(hprev_p), (long)l.K},
                                                                              (long)B * l.K, planes, tf, B, l.N, l.K, st, dbgq);
                    }
                    return (int)launch_x6_tr(X6DenseA{hprev, (long)l.K}, planes, tf, B, l.N, l.K, st, dbgq, ktm);
                }
                // MRL_X6_DBG=1: phase timestamps of workgroup 0 land behind the zero page (scripts/x6_phases.py)
                long long* dbgp = dbg_option("x6_dbg", "MRL_X6_DBG") == 1 ? dbgbuf : nullptr;
#ifdef MRL_X6_EXPERIMENTS
                if (x6_specialised())
                    return (int)launch_gemm_x6s(X6DenseA{hprev, (long)l.K}, planes, ef, B, l.N, l.K, num_cus(), f32_split_mode() == 2, st, dbgp);
#endif
                return (int)launch_gemm_x6(X6DenseA{hprev, (long)l.K}, planes, ef, B, l.N, l.K, st, dbgp,
                                           f32_split_mode() == 2);
            }
            RowKC af{hprev, l.K, B, l.K, is_vec(hprev, l.K), nullptr};
            return gemm_dispatch(l.name, "fwd", var, af, bf, ef, B, l.N, l.K, 1, l.K, st);
        }
    }
}

static int net_forward(const mrl_model* m, const Net& net, const In& in, const float* params, NetWs& nw, int B,
                       hipStream_t st, float* part = nullptr, size_t part_floats = 0) {
    for (size_t i = 0; i < net.L.size(); ++i) {
        // bit masks only where a consumer exists: the layer above is a conv whose data gradient the tiled engine computes
        const bool hb = i < nw.mbits.size() && i + 1 < net.L.size() && net.L[i + 1].kind == 0;
        const bool hpi = i < nw.hp.size();
        const uint16_t* hprev_p = (i && i - 1 < nw.hp.size() && nw.hpvalid[i - 1]) ? nw.hp[i - 1] : nullptr;
        Layer lcopy;
        const Layer* lp = &net.L[i];
        if (lp->ln) { lcopy = *lp; lcopy.act = ACT_NONE; lp = &lcopy; }          // the affine map alone; LN + activation below
        int rc = layer_forward<kExp>(m, *lp, i == 0, in, i ? nw.h[i - 1] : nullptr, params, nw.h[i], nw.planes, nw.dbg, B, st,
                               hb ? nw.mbits[i] : nullptr, hb ? &nw.mvalid[i] : nullptr, hprev_p, hpi ? nw.hp[i] : nullptr,
                               hpi ? &nw.hpvalid[i] : nullptr, part, part_floats);
        if (rc) return rc;
        if (net.L[i].ln) {
            const Layer& l = net.L[i];
            ProfScope ps("layer_norm.fwd", 0.0, 12.0 * B * l.N, st);
            const int blocks = (int)std::min<long>(((long)B + 3) / 4, 2048);
            hipLaunchKernelGGL(ln_fwd_kernel, dim3(blocks), dim3(256), 0, st, nw.h[i], i < nw.xhat.size() ? nw.xhat[i] : nullptr,
                               i < nw.istd.size() ? nw.istd[i] : nullptr, params + l.beta_off, params + l.gamma_off, B, l.N, l.act);
            MRL_LAUNCH_CHECK();
        }
    }
    return 0;
}

static inline const float* hprev_of(const NetWs& nw, int i) { return i ? nw.h[i - 1] : nullptr; }
static int ldsdgrad_kind(const Layer& l, const float* dz) {
    if (l.kind != 0 || layer_padded(l) || (uintptr_t)dz % 16 != 0) return 0;
    if (l.H == 20 && l.W == 20 && l.C == 32 && l.rf == 4 && l.stride == 2 && l.NF == 64) return 1;
    if (l.H == 9 && l.W == 9 && l.C == 64 && l.rf == 3 && l.stride == 1 && l.NF == 64) return 2;
    return 0;
}
static bool tuned(const Layer& l, const char* pass) {
    return tune_table().find(std::string(l.name) + "." + pass) != tune_table().end();
}

// backward through one net; nw.dz[last] already holds dloss/d(pre-activation of the last layer)
template <bool EXP>
static int net_backward(const mrl_model* m, const Net& net, const In& in, const float* params, NetWs& nw, Ws& ws,
                        float* grads, int B, int accumulate, hipStream_t st, StepCtx& ctx, bool is_pi) {
    const bool dzplanes = EXP && (act_planes_mode() & (2 | 32)) != 0;     // 32: planes of the last layer's dz only (conversion pass)
    for (auto& v : nw.dzpvalid) v = 0;
    for (int i = (int)net.L.size() - 1; i >= 0; --i) {
        const Layer& l = net.L[i];
        if (l.ln && ctx.nwstream) {     // the layer-norm reduction below uses ws.part on `st`: wait for the weight gradients in flight
            for (int j = 0; j < ctx.nwstream; ++j) {
                MRL_HIP_CHECK(hipEventRecord(ctx.ev_join, ctx.wstream[j]));
                MRL_HIP_CHECK(hipStreamWaitEvent(st, ctx.ev_join, 0));
            }
        }
        if (l.ln) {
            // nw.dz[i] arrives as the gradient w.r.t. the activation's input = the layer-norm output; turn it into the gradient
            // w.r.t. the affine map's output (in place) and reduce dbeta | dgamma over the rows
            const int blocks = (int)std::min<long>(std::min<long>(((long)B + 3) / 4, LN_MAXBLK), (long)(ws.part_floats / (2 * l.N)));
            if (blocks < 1 || (size_t)i >= nw.xhat.size() || !nw.xhat[i]) return MRL_ENOSPC;
            {
                ProfScope ps("layer_norm.bwd", 0.0, 16.0 * B * l.N, st);
                hipLaunchKernelGGL(ln_bwd_kernel<0>, dim3(blocks), dim3(256), (size_t)8 * l.N * sizeof(float), st, nw.dz[i], nw.xhat[i],
                                   nw.istd[i], params + l.gamma_off, B, l.N, ws.part);
                MRL_LAUNCH_CHECK();
            }
            int rcl = reduce_slabs(ws.part, 2L * l.N, blocks, grads + l.beta_off, 2L * l.N, accumulate, st, &ctx);
            if (rcl) return rcl;
        }
        const float* dz = nw.dz[i];
        // plane tensor of dz[i] (the A operand of this layer's data gradient): written by the data gradient of the layer
        // above, or -- for the last layer, whose dz comes from the heads kernel -- by a conversion pass
        const bool x6dgrad = i > 0 && nw.planes && f32_split_mode() == 2 && !x6_specialised() && !tuned(l, "dgrad");
        if constexpr (EXP) {
        if (dzplanes && x6dgrad && i == (int)net.L.size() - 1 && (size_t)i < nw.dzp.size() && nw.dzp[i] && l.kind == 1 && B >= 1024 &&
            l.K >= 128 && l.N % 32 == 0 && gemm_x6_ok(dz, l.N, l.N)) {
            ProfScope ps("dz.planes", 0.0, 10.0 * B * l.N, st);
            hipError_t e = launch_planes_from_f32(dz, (long)B * l.N, nw.dzp[i], (long)B * l.N, st);
            if (e != hipSuccess) return (int)e;
            nw.dzpvalid[i] = 1;
        }
        }
        const uint16_t* dz_p = (dzplanes && x6dgrad && (size_t)i < nw.dzp.size() && nw.dzpvalid[i]) ? nw.dzp[i] : nullptr;
        // where the data gradient may leave the plane tensor of dz[i-1]
        uint16_t* dx_p = (EXP && (act_planes_mode() & 2) && x6dgrad && i >= 2 && (size_t)i - 1 < nw.dzp.size() && net.L[i - 1].act == ACT_RELU) ? nw.dzp[i - 1] : nullptr;
        const float* hmask = hprev_of(nw, i);                        // act' source of the layer below
        const float* hprev = i ? nw.h[i - 1] : nullptr;
        const long rows = layer_rows(l, B);
        const bool first = (i == 0);
        if (rows > 0x7fffffffL || l.b_off != l.w_off + (long)l.K * l.N) return MRL_EUNSUP;
        // ---- weight + bias gradient: dW[k][n] = sum_rows A[row][k] * dz[row][n], db[n] = sum_rows dz[row][n]
        //      (split-K over rows; the bias column sums ride on the B operand's LDS image)
        hipStream_t stw = st;
        float* wpart = ws.part;
        if (ctx.nwstream) {
            const int j = ctx.wrr++ % ctx.nwstream;
            MRL_HIP_CHECK(hipEventRecord(ctx.ev_fork, st));
            MRL_HIP_CHECK(hipStreamWaitEvent(ctx.wstream[j], ctx.ev_fork, 0));
            stw = ctx.wstream[j];
            if (ctx.wpart[j]) wpart = ctx.wpart[j];
        }
        int var = pick_variant(l.name, "wgrad", l.K, l.N);
        const long slab = (long)l.K * l.N + l.N;
        const void* asrc = first ? in.obs : (const void*)hprev;
        const int ik = imgres_kind(l, first && m->d.ob_dtype == MRL_OB_U8, asrc);
        int rc;
        // conv2 / conv3 / fc1 at training batch sizes: eight exact bf16 products per multiply on the bf16 pipe
        // (wgradx8.hip.h), the arithmetic mode of the forward / data-gradient engines
        WgradX8Plan wx;
        if (!first && f32_split_mode() == 2 && get_option("wgrad_x8", "MRL_WGRAD_X8", 1) && !tuned(l, "wgrad") && !layer_padded(l) &&
            (uintptr_t)hprev % 16 == 0 && (uintptr_t)dz % 16 == 0 && (l.kind != 0 || ((l.rf * l.C) % 4 == 0 && l.C % 4 == 0)))
            wx = wgrad_x8_plan(rows, l.K, l.N, num_cus(), ws.part_floats, get_option("wgrad_x8", "MRL_WGRAD_X8", 1) >= 2);
        // conv2 / conv3 of NatureCNN: image-resident kernel, eight exact bf16 products per multiply, transpose reads
        // (wgradtr.hip.h); option wgrad_tr = 0 keeps the fp32-MFMA image-resident engine (imgres.hip.h)
        const bool trw = !first && !wx.cfg && (ik == 2 || ik == 3) && f32_split_mode() == 2 && !tuned(l, "wgrad") &&
                         (uintptr_t)dz % 16 == 0 && get_option("wgrad_tr", "MRL_WGRAD_TR", 1);
        // fully connected layers at training batch sizes: natural-layout staging + transpose reads (wgradtr.hip.h, dense form)
        WgTrDensePlan wd;
        if (!first && l.kind == 1 && wx.cfg && get_option("wgrad_tr", "MRL_WGRAD_TR", 1) == 1)
            wd = wgrad_tr_dense_plan(rows, l.K, l.N, l.K, num_cus(), ws.part_floats);
        if (wd.mt) {
            {
                char label[40];
                if (prof_enabled()) snprintf(label, sizeof label, "%s.wgrad", l.name);
                ProfScope ps(label, 2.0 * rows * (double)l.K * l.N, 0.0, stw);
                hipError_t e = launch_wgrad_tr_dense(hprev, l.K, dz, wpart, slab, (int)rows, l.K, l.N, wd, stw);
                if (e != hipSuccess) return (int)e;
            }
            rc = reduce_slabs(wpart, slab, wd.nslab, grads + l.w_off, slab, accumulate, stw, &ctx);
            if (rc) return rc;
        } else if (trw) {
            int nblocks = (int)std::min<long>(std::min<long>(num_cus(), IMGRES_MAX_BLOCKS), B);
            nblocks = (int)std::min<long>(nblocks, (long)(ws.part_floats / slab));
            if (nblocks < 1) return MRL_ENOSPC;
            {
                char label[40];
                if (prof_enabled()) snprintf(label, sizeof label, "%s.wgrad", l.name);
                ProfScope ps(label, 2.0 * rows * (double)l.K * l.N, 0.0, stw);
                const int wtv = get_option("wgrad_tr", "MRL_WGRAD_TR", 1);
                hipError_t e;
#define MRL_WT(XP2, DP2, XP3, DP3, DBG)                                                                                   \
    (ik == 2 ? launch_wgrad_tr<20, 20, 32, 4, 2, 64, 8, 2, 2, XP2, DP2, DBG>(hprev, dz, B, wpart, nblocks, stw)       \
             : launch_wgrad_tr<9, 9, 64, 3, 1, 64, 12, 3, 1, XP3, DP3, DBG>(hprev, dz, B, wpart, nblocks, stw))
                switch (wtv) {
#ifdef MRL_X6_EXPERIMENTS       // 2-4: timing experiments (unpadded pixel strides; staging / MFMA phase left out)
                case 2: e = MRL_WT(0, 0, 0, 0, 0); break;
                case 3: e = MRL_WT(32, 64, 64, 64, 1); break;
                case 4: e = MRL_WT(32, 64, 64, 64, 2); break;
#endif
                default: e = MRL_WT(32, 64, 64, 64, 0); break;
                }
#undef MRL_WT
                if (e != hipSuccess) return (int)e;
            }
            rc = reduce_slabs(wpart, slab, nblocks, grads + l.w_off, slab, accumulate, stw, &ctx);
            if (rc) return rc;
        } else if (wx.cfg) {
            char label[40];
            if (prof_enabled()) snprintf(label, sizeof label, "%s.wgrad", l.name);
            hipError_t e;
            {
                ProfScope ps(label, 2.0 * rows * (double)l.K * l.N, 0.0, stw);
                if (l.kind == 0) {
                    X6ConvA ca;
                    fill_conv(ca, l, hprev, (int)rows, nullptr);
                    e = launch_wgrad_x8(ca, dz, wpart, slab, (int)rows, l.K, l.N, wx, stw);
                } else {
                    e = launch_wgrad_x8(X6DenseA{hprev, (long)l.K}, dz, wpart, slab, (int)rows, l.K, l.N, wx, stw);
                }
            }
            if (e != hipSuccess) return (int)e;
            rc = reduce_slabs(wpart, slab, wx.nslab, grads + l.w_off, slab, accumulate, stw, &ctx);
            if (rc) return rc;
        } else if (ik == 1 && first && !tuned(l, "wgrad") && get_option("u8_bf16x3", "MRL_U8_BF16X3", 1) &&
                   get_option("c1_wgrad2", "MRL_C1_WGRAD2", 3) >= 2) {
            // conv1: half-image work units, two 4-wave workgroups per CU (c1wgrad.hip.h)
            int nblocks = (int)std::min<long>(std::min<long>(2L * num_cus(), 2L * B), (long)(ws.part_floats / slab));
            if (nblocks < 1) return MRL_ENOSPC;
            {
                char label[40];
                if (prof_enabled()) snprintf(label, sizeof label, "%s.wgrad", l.name);
                ProfScope ps(label, 2.0 * B * l.OH * l.OW * (double)l.K * l.NF, 0.0, stw);
                hipError_t e = launch_c1wgrad_half(asrc, in.srow, dz, B, wpart, nblocks, stw, get_option("c1_wgrad2", "MRL_C1_WGRAD2", 3) >= 3,
                                                   std::max(0, dbg_option("c1_dbg", "MRL_C1_DBG") - 64));
                if (e != hipSuccess) return (int)e;
            }
            rc = reduce_slabs(wpart, slab, nblocks, grads + l.w_off, slab, accumulate, stw, &ctx);
            if (rc) return rc;
        } else if (ik && (var == V_IMGRES || tune_table().find(std::string(l.name) + ".wgrad") == tune_table().end())) {
            int nblocks = (int)std::min<long>(std::min<long>(num_cus(), IMGRES_MAX_BLOCKS), B);
            nblocks = (int)std::min<long>(nblocks, (long)(ws.part_floats / slab));
            if (nblocks < 1) return MRL_ENOSPC;
            rc = imgres_dispatch(ik, l, asrc, first ? in.srow : nullptr, dz, nullptr, B, wpart, nblocks, stw);
            if (rc) return rc;
            rc = reduce_slabs(wpart, slab, nblocks, grads + l.w_off, slab, accumulate, stw, &ctx);
            if (rc) return rc;
        } else {
        if (var >= V_WRES16) var = l.N <= 32 ? V_128x32 : (l.N <= 64 ? V_128x64_W41 : V_128x128);
        // ---- weight + bias gradient: dW[k][n] = sum_rows A[row][k] * dz[row][n], db[n] = sum_rows dz[row][n]
        //      (split-K over rows; the bias column sums ride on the B operand's LDS image)
        const Split sp = pick_split(var, l.K, l.N, rows);
        if ((size_t)sp.nsplit * slab > ws.part_floats) return MRL_ENOSPC;
        EpiPartial ep{wpart, slab, l.N, (long)l.K * l.N};
        RowMC bfm{dz, l.N, l.N, (int)rows, is_vec(dz, l.N), nullptr};
        if (l.kind == 0) {
            if (first && m->d.ob_dtype != MRL_OB_U8) {
                ConvPatchMC<2> af;
                fill_conv(af, l, in.obs, (int)rows, in.srow);
                rc = gemm_dispatch(l.name, "wgrad", var, af, bfm, ep, l.K, l.N, (int)rows, sp.nsplit, sp.ksplit, stw);
            } else if (first) {
                ConvPatchMC<1> af;
                fill_conv(af, l, in.obs, (int)rows, in.srow);
                rc = gemm_dispatch(l.name, "wgrad", var, af, bfm, ep, l.K, l.N, (int)rows, sp.nsplit, sp.ksplit, stw);
            } else {
                ConvPatchMC<false> af;
                fill_conv(af, l, hprev, (int)rows, nullptr);
                rc = gemm_dispatch(l.name, "wgrad", var, af, bfm, ep, l.K, l.N, (int)rows, sp.nsplit, sp.ksplit, stw);
            }
        } else {
            if (first) {
                RowMC af{(const float*)in.obs, l.K, l.K, B, is_vec(in.obs, l.K), in.srow};
                rc = gemm_dispatch(l.name, "wgrad", var, af, bfm, ep, l.K, l.N, B, sp.nsplit, sp.ksplit, stw);
            } else {
                RowMC af{hprev, l.K, l.K, B, is_vec(hprev, l.K), nullptr};
                rc = gemm_dispatch(l.name, "wgrad", var, af, bfm, ep, l.K, l.N, B, sp.nsplit, sp.ksplit, stw);
            }
        }
        if (rc) return rc;
        rc = reduce_slabs(wpart, slab, sp.nsplit, grads + l.w_off, slab, accumulate, stw, &ctx);
        if (rc) return rc;
        }
        // data parallel: the tail of the flat gradient [this layer .. heads] is final -> it travels (RCCL, communication
        // stream) while the layers below are still being back-propagated
        if (ctx.comm && ctx.final_chunk && is_pi && i == ctx.early_layer) {
            rc = comm_allreduce_async(ctx.comm, grads + ctx.early_lo, m->P - ctx.early_lo, ctx.rank_weight, st);
            if (rc) return rc;
        }
        // ---- data gradient into dz[i-1] (masked by act' of layer i-1)
        if (!first) {
            const Layer& lp = net.L[i - 1];
            if (l.kind == 0) {
                DgradGeom g;
                g.H = l.H; g.W = l.W; g.C = l.C; g.rf = l.rf; g.stride = l.stride; g.OH = l.OH; g.OW = l.OW;
                if (layer_padded(l)) {      // enumerate the zero-padded input; the epilogue maps back and drops the border
                    g.Hr = l.H; g.Wr = l.W; g.pad_t = l.pad_t; g.pad_l = l.pad_l;
                    g.H = (l.OH - 1) * l.stride + l.rf; g.W = (l.OW - 1) * l.stride + l.rf;
                }
                g.NF = l.NF; g.taps = (l.rf + l.stride - 1) / l.stride;
                g.HY = (g.H + l.stride - 1) / l.stride; g.WX = (g.W + l.stride - 1) / l.stride; g.B = B;
                g.finish();
                if (l.NF % 4 != 0 || (long)B * g.HY * g.WX > 0x7fffffffL) return MRL_EUNSUP;
                int Kd = g.taps * g.taps * l.NF;
                int Md = B * g.HY * g.WX;
                const double fl = 2.0 * B * l.OH * l.OW * (double)l.K * l.NF;
                const bool wok = wres_dgrad_ok(l);
                // LDS-resident engine for the NatureCNN geometries (ldsdgrad.hip.h is templated on the shape)
                const int lk = ldsdgrad_kind(l, dz);
                const bool overridden = tune_table().find(std::string(l.name) + ".dgrad") != tune_table().end();
                int dv = pick_variant(l.name, "dgrad", Md, l.C, false);
                if (lk && !overridden && nw.planes && f32_split_mode() && get_option("dgrad_x6", "MRL_DGRAD_X6", 2)) {
                    dgrad_x6_pipe() = get_option("dgrad_x6", "MRL_DGRAD_X6", 2) >= 2;
                    // position-major tiles on the split-bf16 pipe (dgradx6.hip.h): only useful MACs, all parity classes in one GEMM
                    char label[40];
                    if (prof_enabled()) snprintf(label, sizeof label, "%s.dgrad", l.name);
                    ProfScope ps(label, fl, 0.0, st);
                    const bool x8 = f32_split_mode() == 2;
                    const int dbg = dbg_option("dgx6_dbg", "MRL_DGX6_DBG");
                    // act' of the layer below: its ReLU bit mask when this call's forward pass wrote one, else its fp32 output
                    const uint32_t* bits = ((size_t)i - 1 < nw.mvalid.size() && nw.mvalid[i - 1] && lp.act == ACT_RELU) ? nw.mbits[i - 1] : nullptr;
                    hipError_t e;
#ifdef MRL_X6_EXPERIMENTS
                    if (x6_specialised() && !dbg)
                        e = lk == 1 ? launch_dgrad_x6s<20, 20, 32, 4, 2, 64, 2, 2>(dz, params + l.w_off, hmask, bits, nw.dz[i - 1], lp.act, B, nw.planes, x8, num_cus(), st)
                                    : launch_dgrad_x6s<9, 9, 64, 3, 1, 64, 4, 1>(dz, params + l.w_off, hmask, bits, nw.dz[i - 1], lp.act, B, nw.planes, x8, num_cus(), st);
                    else
#endif
                    {
                        const bool trpl = (act_planes_mode() & 8) != 0;
                        e = lk == 1 ? launch_dgrad_x6<20, 20, 32, 4, 2, 64, 2, 2, EXP>(dz, params + l.w_off, hmask, bits, nw.dz[i - 1], lp.act, B, nw.planes, x8, num_cus(), st, dbg, dz_p, dx_p, trpl)
                                    : launch_dgrad_x6<9, 9, 64, 3, 1, 64, 4, 1, EXP>(dz, params + l.w_off, hmask, bits, nw.dz[i - 1], lp.act, B, nw.planes, x8, num_cus(), st, dbg, dz_p, dx_p, trpl);
                        if (dx_p && x8 && !dbg) nw.dzpvalid[i - 1] = 1;
                    }
                    rc = (int)e;
                } else
                if (lk && (dv == V_LDSDGRAD || !overridden)) {
                    char label[40];
                    if (prof_enabled()) snprintf(label, sizeof label, "%s.dgrad", l.name);
                    ProfScope ps(label, fl, 0.0, st);
                    hipError_t e;
                    const float* wsrc = params + l.w_off;
                    // MRL_DGRAD_DBG=<layer index>: phase timestamps of workgroup 0 land behind the zero page
                    long long* dbgp = dbg_option("dgrad_dbg", "MRL_DGRAD_DBG") == i ? reinterpret_cast<long long*>(ws.zeros) + 64 : nullptr;
                    // conv2 (4 taps per class: the memory phases are a third of a group): asynchronous variant (LDS-DMA
                    // staging, mask prefetch) 9.05 -> 8.7 ms; conv3 (9 taps): 6.6 -> 6.9 ms, stays on the synchronous kernel
                    // (16 waves x 1 row tile, groups of 6 images; 8 x 2 and two half-size workgroups per CU are slower)
                    if (lk == 1 && get_option("dgrad_async", "MRL_DGRAD_ASYNC", 1))
                        e = launch_lds_dgrad_async<20, 20, 32, 4, 2, 64, 5, 8, 2>(dz, wsrc, hmask, nw.dz[i - 1], lp.act, B, num_cus(), dbgp, st);
                    else if (lk == 1)
                        e = launch_lds_dgrad<20, 20, 32, 4, 2, 64, 5, 16, 1>(dz, wsrc, hmask, nw.dz[i - 1], lp.act, B, num_cus(), dbgp, st);
                    else
                        e = launch_lds_dgrad<9, 9, 64, 3, 1, 64, 6, 16, 1>(dz, wsrc, hmask, nw.dz[i - 1], lp.act, B, num_cus(), dbgp, st);
                    rc = (int)e;
                } else
                if (dv >= V_WRES16 && wok) {
                    const int zc = l.stride * l.stride;
                    const long tpc = ((long)Md + 31) / 32;
                    WresDgradA wa; static_cast<DgradGeom&>(wa) = g; wa.dz = dz; wa.zeros = ws.zeros; wa.tiles_per_class = tpc;
                    WresDgradB wb; static_cast<DgradGeom&>(wb) = g; wb.w = params + l.w_off;
                    WresEpiDgrad we; static_cast<DgradGeom&>(we) = g; we.out = nw.dz[i - 1]; we.hprev = hprev;
                    we.act = lp.act; we.tiles_per_class = tpc;
                    rc = wres_dispatch(l.name, "dgrad", dv, l.C, wa, wb, we, zc, Kd, tpc * zc, fl, st);
                } else if (get_option("conv_skinny", "MRL_CONV_SKINNY", 1) && !overridden && l.NF % 8 == 0 && l.C % 32 == 0 &&
                           (uintptr_t)dz % 16 == 0 && (uintptr_t)(params + l.w_off) % 8 == 0 &&
                           ((long)Md + 127) / 128 * (l.stride * l.stride) <= 2L * num_cus()) {
                    // latency-bound sizes: register-direct skinny tiles (convskinny.hip.h)
                    char label[40];
                    if (prof_enabled()) snprintf(label, sizeof label, "%s.dgrad", l.name);
                    ProfScope ps(label, fl, 0.0, st);
                    EpiDgradConv ef; static_cast<DgradGeom&>(ef) = g; ef.out = nw.dz[i - 1]; ef.h = hmask; ef.act = lp.act;
                    rc = (int)launch_conv_skinny_dgrad(g, dz, params + l.w_off, ef, st);
                } else {
                    if (dv >= V_WRES16) dv = l.C <= 32 ? V_128x32 : V_128x64_W41;
                    DgradA af; static_cast<DgradGeom&>(af) = g; af.dz = dz;
                    DgradB bf; static_cast<DgradGeom&>(bf) = g; bf.w = params + l.w_off;
                    EpiDgradConv ef; static_cast<DgradGeom&>(ef) = g; ef.out = nw.dz[i - 1]; ef.h = hmask; ef.act = lp.act;
                    rc = gemm_dispatch(l.name, "dgrad", dv, af, bf, ef, Md, l.C, Kd, l.stride * l.stride, Kd, st, fl);
                }
            } else if (nw.planes && B >= 1024 && l.K >= 128 && gemm_x6_ok(dz, l.N, l.N) && !tuned(l, "dgrad") &&
                       f32_split_mode()) {
                // dX[b][k] = sum_n dz[b][n] W[k][n]: W's rows are already the k-contiguous B operand
                char label[40];
                if (prof_enabled()) snprintf(label, sizeof label, "%s.dgrad", l.name);
                ProfScope ps(label, 2.0 * B * (double)l.K * l.N, 0.0, st);
                EpiMaskAct ef{nw.dz[i - 1], l.K, hmask, lp.act};
                const bool tr = (dx_p || ((act_planes_mode() & 8) && f32_split_mode() == 2 && !x6_specialised() && lp.act == ACT_RELU)) &&
                                l.K % 32 == 0 && (uintptr_t)hmask % 16 == 0;
                const bool ktm = tr && !dz_p && x6_ktm() && x6_tr_staged(l.K, nullptr);
                hipError_t e = launch_split_planes(params + l.w_off, l.K, l.N, false, nw.planes, st, dz_p != nullptr, 0, 1, 32, ktm);
                if (e == hipSuccess && (dz_p || tr)) {
                    // transposed-accumulator epilogue; experiment builds: dz staged from its plane tensor, dz[i-1] leaves with
                    // its planes (planes.hip.h)
                    const X6DenseA da{dz_p ? reinterpret_cast<const float*>(dz_p) : dz, (long)l.N};
                    const long aps = (long)B * l.N;
                    if (tr) {
                        TrMaskRelu tf{nw.dz[i - 1], l.K, hmask, nullptr, dx_p, (long)B * l.K};
                        bool done = false;
                        if constexpr (EXP) {
                            if (dz_p) { e = launch_gemm_x6_planes<true, true>(da, aps, nw.planes, tf, B, l.K, l.N, st); done = true; }
                        }
                        if (!done) e = launch_x6_tr(da, nw.planes, tf, B, l.K, l.N, st, nullptr, ktm);
                        if (e == hipSuccess && dx_p) nw.dzpvalid[i - 1] = 1;
                    } else {
                        if constexpr (EXP) e = launch_gemm_x6_planes<true, false>(da, aps, nw.planes, ef, B, l.K, l.N, st);
                    }
                } else if (e == hipSuccess) {
#ifdef MRL_X6_EXPERIMENTS
                    if (x6_specialised())
                        e = launch_gemm_x6s(X6DenseA{dz, (long)l.N}, nw.planes, ef, B, l.K, l.N, num_cus(), f32_split_mode() == 2, st);
                    else
#endif
                    e = launch_gemm_x6(X6DenseA{dz, (long)l.N}, nw.planes, ef, B, l.K, l.N, st, nullptr, f32_split_mode() == 2);
                }
                rc = (int)e;
            } else {
                RowKC af{dz, l.N, B, l.N, is_vec(dz, l.N), nullptr};
                const float* W = params + l.w_off;
                RowKC bf{W, l.N, l.K, l.N, is_vec(W, l.N), nullptr};
                EpiMaskAct ef{nw.dz[i - 1], l.K, hmask, lp.act};
                int dv = pick_variant(l.name, "dgrad", B, l.K);
                rc = gemm_dispatch(l.name, "dgrad", dv, af, bf, ef, B, l.K, l.N, 1, l.N, st);
            }
            if (rc) return rc;
        }
    }
    return 0;
}


// ============================================================================================
// recurrent cell: x@wx GEMM + scans (lstm.hip.h) + weight / input gradients as GEMMs over all steps
// ============================================================================================
static int lstm_forward(const Net& net, const In& in, const float* params, NetWs& nw, int nseq, int T, const float* s0,
                        const uint8_t* mask, const int32_t* msrow, float* s_out, bool train, hipStream_t st) {
    const int B = nseq * T, nh = net.nh, N4 = 4 * nh, K = net.lstm_nin;
    const float* wx = params + net.wx_off;
    RowMC bf{wx, N4, N4, K, is_vec(wx, N4), nullptr};
    EpiMaskAct ef{nw.zx, N4, nullptr, ACT_NONE};                 // plain product: the bias joins inside the scan
    const int var = pick_variant("lstm", "fwd", B, N4);
    int rc;
    if (net.L.empty()) {                                          // tf.layers.flatten(X): observations, gathered in place
        RowKC af{(const float*)in.obs, K, B, K, is_vec(in.obs, K), in.srow};
        rc = gemm_dispatch("lstm", "fwd", var, af, bf, ef, B, N4, K, 1, K, st);
    } else {
        const float* x = nw.h.back();
        RowKC af{x, K, B, K, is_vec(x, K), nullptr};
        rc = gemm_dispatch("lstm", "fwd", var, af, bf, ef, B, N4, K, 1, K, st);
    }
    if (rc) return rc;
    LstmFwdArgs a;
    a.zx = nw.zx; a.wh = params + net.wh_off; a.bias = params + net.lb_off; a.s0 = s0; a.mask = mask; a.srow = msrow;
    a.nenv = nseq; a.T = T;
    a.gates = train ? nw.gates : nullptr; a.cm = train ? nw.cm : nullptr; a.hm = train ? nw.hm : nullptr;
    a.tc = train ? nw.tc : nullptr; a.hout = nw.hout; a.s_out = s_out;
    if (net.lnl) {
        {   // zx <- LN(x@wx) * gx + bx, row-parallel (a2c/utils.py:130, first term)
            ProfScope ps("lnlstm.ln_x", 0.0, 12.0 * B * N4, st);
            const int blocks = (int)std::min<long>(((long)B + 3) / 4, 2048);
            hipLaunchKernelGGL(ln_fwd_kernel, dim3(blocks), dim3(256), 0, st, nw.zx, train ? nw.xhx : nullptr, train ? nw.isx : nullptr,
                               params + net.gx_off + N4, params + net.gx_off, B, N4, ACT_NONE, LNLSTM_EPS);
            MRL_LAUNCH_CHECK();
        }
        LnLstmFwdArgs p;
        p.base = a;
        p.gh = params + net.gh_off; p.bh = p.gh + N4; p.gc = params + net.gc_off; p.bc = p.gc + nh;
        p.xhh = train ? nw.xhh : nullptr; p.ish = train ? nw.ish : nullptr;
        p.xhc = train ? nw.xhc : nullptr; p.isc = train ? nw.isc : nullptr;
        ProfScope ps("lnlstm.scan_fwd", 2.0 * B * (double)nh * N4, 0.0, st);
        return (int)launch_lnlstm_fwd(p, nh, num_cus(), st);
    }
    ProfScope ps("lstm.scan_fwd", 2.0 * B * (double)nh * N4, 0.0, st);
    return (int)launch_lstm_fwd(a, nh, num_cus(), st);
}

// nw.dhout holds dL/dh_t (written by the heads kernel).  Leaves dL/d(features) in nw.dz.back() (cnn_lstm).
static int lstm_backward(const Net& net, const In& in, const float* params, NetWs& nw, Ws& ws, float* grads, int nseq, int T,
                         const uint8_t* mask, const int32_t* msrow, hipStream_t st, StepCtx& ctx) {
    const int B = nseq * T, nh = net.nh, N4 = 4 * nh, K = net.lstm_nin;
    {
        LstmBwdArgs a;
        a.dhout = nw.dhout; a.wh = params + net.wh_off; a.gates = nw.gates; a.cm = nw.cm; a.tc = nw.tc; a.mask = mask;
        a.srow = msrow; a.nenv = nseq; a.T = T; a.dzg = nw.dzg;
        if (net.lnl) {
            LnLstmBwdArgs p;
            p.base = a;
            p.gh = params + net.gh_off; p.gc = params + net.gc_off;
            p.xhh = nw.xhh; p.ish = nw.ish; p.xhc = nw.xhc; p.isc = nw.isc; p.dzh = nw.dzh; p.dcn = nw.dcn;
            {
                ProfScope ps("lnlstm.scan_bwd", 2.0 * B * (double)nh * N4, 0.0, st);
                hipError_t e = launch_lnlstm_bwd(p, nh, num_cus(), st);
                if (e != hipSuccess) return (int)e;
            }
            // gains and shifts of the three LNs + b, as column sums over all B rows (fixed row order per block, slabs reduced
            // in fixed order); the x path's LN Jacobian turns dzg into the gradient w.r.t. the raw x@wx in place
            auto cols = [&](int mode, float* dy, const float* xhat, const float* istd, const float* gain, int N, long out_off,
                            long also_shift_to) -> int {
                const int blocks = (int)std::min<long>(std::min<long>(((long)B + 3) / 4, LN_MAXBLK), (long)(ws.part_floats / (2 * N)));
                if (blocks < 1) return MRL_ENOSPC;
                ProfScope ps("lnlstm.ln_bwd", 0.0, 16.0 * B * N, st);
                if (mode == 1)
                    hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(blocks), dim3(256), (size_t)8 * N * sizeof(float), st, dy, xhat, istd, gain,
                                       B, N, ws.part);
                else
                    hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(blocks), dim3(256), (size_t)8 * N * sizeof(float), st, dy, xhat, istd, gain,
                                       B, N, ws.part);
                MRL_LAUNCH_CHECK();
                int rc = reduce_slabs(ws.part, 2L * N, blocks, grads + out_off, 2L * N, 0, st, &ctx);
                if (!rc && also_shift_to >= 0) rc = reduce_slabs(ws.part + N, 2L * N, blocks, grads + also_shift_to, N, 0, st, &ctx);
                return rc;
            };
            int rc;
            if ((rc = cols(2, nw.dcn, nw.xhc, nullptr, nullptr, nh, net.gc_off, -1))) return rc;
            if ((rc = cols(2, nw.dzg, nw.xhh, nullptr, nullptr, N4, net.gh_off, net.lb_off))) return rc;       // db = dbh = sum dz
            if ((rc = cols(1, nw.dzg, nw.xhx, nw.isx, params + net.gx_off, N4, net.gx_off, -1))) return rc;
        } else {
            ProfScope ps("lstm.scan_bwd", 2.0 * B * (double)nh * N4, 0.0, st);
            hipError_t e = launch_lstm_bwd(a, nh, num_cus(), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    RowMC bfm{nw.dzg, N4, N4, B, is_vec(nw.dzg, N4), nullptr};
    // dwx[k][n] = sum_b x[b][k] dz[b][n],  db[n] = sum_b dz[b][n] (column sums ride on the B operand)
    {
        int var = pick_variant("lstm", "wgrad", K, N4);
        if (var >= V_WRES16) var = V_128x128;
        const long slab = (long)K * N4 + N4;
        const Split sp = pick_split(var, K, N4, B);
        if ((size_t)sp.nsplit * slab > ws.part_floats) return MRL_ENOSPC;
        EpiPartial ep{ws.part, slab, N4, (long)K * N4};
        int rc;
        if (net.L.empty()) {
            RowMC af{(const float*)in.obs, K, K, B, is_vec(in.obs, K), in.srow};
            rc = gemm_dispatch("lstm", "wgrad", var, af, bfm, ep, K, N4, B, sp.nsplit, sp.ksplit, st);
        } else {
            const float* x = nw.h.back();
            RowMC af{x, K, K, B, is_vec(x, K), nullptr};
            rc = gemm_dispatch("lstm", "wgrad", var, af, bfm, ep, K, N4, B, sp.nsplit, sp.ksplit, st);
        }
        if (rc) return rc;
        if ((rc = reduce_slabs(ws.part, slab, sp.nsplit, grads + net.wx_off, (long)K * N4, 0, st, &ctx))) return rc;
        // (layer-normalised cell: nw.dzg is the x path's raw gradient here, and b already has its own column sums)
        if (!net.lnl && (rc = reduce_slabs(ws.part + (long)K * N4, slab, sp.nsplit, grads + net.lb_off, N4, 0, st, &ctx))) return rc;
    }
    // dwh[k][n] = sum_b hm[b][k] dz[b][n]   (hm: the masked previous h the cell actually multiplied)
    {
        int var = pick_variant("lstm_h", "wgrad", nh, N4);
        if (var >= V_WRES16) var = V_128x128;
        const long slab = (long)nh * N4 + N4;
        const Split sp = pick_split(var, nh, N4, B);
        if ((size_t)sp.nsplit * slab > ws.part_floats) return MRL_ENOSPC;
        EpiPartial ep{ws.part, slab, N4, (long)nh * N4};
        RowMC af{nw.hm, nh, nh, B, is_vec(nw.hm, nh), nullptr};
        const float* dzh = net.lnl ? nw.dzh : nw.dzg;                    // gradient w.r.t. the raw h@wh
        RowMC bfh{dzh, N4, N4, B, is_vec(dzh, N4), nullptr};
        int rc = gemm_dispatch("lstm_h", "wgrad", var, af, bfh, ep, nh, N4, B, sp.nsplit, sp.ksplit, st);
        if (rc) return rc;
        if ((rc = reduce_slabs(ws.part, slab, sp.nsplit, grads + net.wh_off, (long)nh * N4, 0, st, &ctx))) return rc;
    }
    // input gradient into the feature net: dfeat[b][k] = (sum_n dz[b][n] wx[k][n]) * act'(feat)
    if (!net.L.empty()) {
        const float* wx = params + net.wx_off;
        RowKC af{nw.dzg, N4, B, N4, is_vec(nw.dzg, N4), nullptr};
        RowKC bf{wx, N4, K, N4, is_vec(wx, N4), nullptr};
        EpiMaskAct ef{nw.dz.back(), K, nw.h.back(), net.L.back().act};
        const int dv = pick_variant("lstm", "dgrad", B, K);
        int rc = gemm_dispatch("lstm", "dgrad", dv, af, bf, ef, B, K, N4, 1, N4, st);
        if (rc) return rc;
    }
    return 0;
}

static void fill_head_args(const mrl_model* m, const float* params, const Ws& ws, HeadArgs& a) {
    memset(&a, 0, sizeof a);
    const Net& pn = m->pi;
    a.nlat = pn.nlat; a.lat_act = pn.lat_act;
    a.lat = ws.pi.lat();
    if (m->vf_copy) {
        a.shared = 0; a.nlatv = m->vf.nlat; a.vlat_act = m->vf.lat_act; a.vlat = ws.vf.h.back();
    } else {
        a.shared = 1; a.nlatv = pn.nlat; a.vlat_act = pn.lat_act; a.vlat = a.lat;
    }
    a.has_pi_head = m->has_pi_head;
    a.Wpi = m->has_pi_head ? params + m->pi_w : nullptr;
    a.bpi = m->has_pi_head ? params + m->pi_b : nullptr;
    a.logstd = m->logstd >= 0 ? params + m->logstd : nullptr;
    a.Wvf = params + m->vf_w; a.bvf = params + m->vf_b;
    a.pd_kind = m->d.pd_kind; a.nact = m->d.nact; a.HP = m->HP;
    a.nsub = m->d.nsub;
    for (int i = 0; i < 16; ++i) a.nvec[i] = m->d.nvec[i];
}

static bool wave_heads_shape(const HeadArgs& a) {
    return a.has_pi_head && a.shared && a.pd_kind == MRL_PD_CATEGORICAL && a.nact <= 8 && a.nlat == 512;
}
static int pick_ts(HeadArgs& a, bool train) {
    HeadLds L;
    for (int ts : {32, 16, 8, 4}) {
        a.TS = ts;
        if (head_lds_carve(a, train, nullptr, L) <= 150 * 1024) return 0;
    }
    return MRL_EUNSUP;
}

// ============================================================================================
// public entry points
// ============================================================================================
static int model_act_impl(const mrl_model* m, const float* params, const void* obs, const float* noise, int n,
                          const float* state_in, const uint8_t* mask, float* state_out,
                          void* actions_out, float* values_out, float* neglogp_out, float* pdparam_out,
                          void* workspace, size_t workspace_bytes, int chunk, void* stream) {
    if (!m || !params || !obs || n <= 0 || chunk <= 0 || !workspace) return MRL_EINVAL;
    if (m->pi.lstm && (!mask || !state_out)) return MRL_EINVAL;      // recurrent policies: mrl_model_act_rnn
    if ((actions_out != nullptr) != (neglogp_out != nullptr)) return MRL_EINVAL;
    if (actions_out && !noise) return MRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    Ws ws;
    carve(m, chunk, (char*)workspace, ws);
    if (ws.total > workspace_bytes) return MRL_ENOSPC;
    const size_t ob_bytes = (size_t)m->ob_elems * (m->d.ob_dtype == MRL_OB_U8 ? 1 : 4);
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int Bc = std::min(chunk, n - c0);
        In in{(const char*)obs + (size_t)c0 * ob_bytes, nullptr};
        int rc = 0;
        // the 2 x 64 tanh MLP on float observations: both layers of both nets in one launch (mlpact.hip.h; bit-identical latents)
        const mrl_model_desc& d = m->d;
        const bool mlp_act = get_option("mlp_act", "MRL_MLP_ACT", 1) && d.network == MRL_NET_MLP && d.num_layers == 2 && !d.layer_norm &&
                             d.num_hidden == 64 && d.activation == MRL_ACT_TANH && d.ob_dtype != MRL_OB_U8 && !m->pi.lstm &&
                             m->pi.L.size() == 2 && (!m->vf_copy || m->vf.L.size() == 2) && m->pi.L[0].kind == 1 &&
                             mlp_act_lds_bytes((int)m->ob_elems, 4, 2) <= 64 * 1024;
        if (mlp_act) {
            MlpActArgs ma;
            ma.params = params; ma.obs = (const float*)in.obs; ma.K0 = (int)m->ob_elems; ma.n = Bc; ma.nets = m->vf_copy ? 2 : 1;
            const Net* nets2[2] = {&m->pi, m->vf_copy ? &m->vf : &m->pi};
            for (int q = 0; q < 2; ++q) {
                ma.w0[q] = nets2[q]->L[0].w_off; ma.b0[q] = nets2[q]->L[0].b_off;
                ma.w1[q] = nets2[q]->L[1].w_off; ma.b1[q] = nets2[q]->L[1].b_off;
            }
            ma.lat[0] = ws.pi.h.back(); ma.lat[1] = m->vf_copy ? ws.vf.h.back() : ws.pi.h.back();
            ProfScope ps("mlp_act", 2.0 * Bc * ma.nets * (double)(ma.K0 * 64 + 64 * 64), 0.0, st);
            hipError_t e = launch_mlp_act_latent(ma, st);
            if (e != hipSuccess) return (int)e;
        } else {
        rc = net_forward(m, m->pi, in, params, ws.pi, Bc, st, ws.part, ws.part_floats);       // (split-K fc layers at act batch sizes)
        if (rc) return rc;
        if (m->pi.lstm) {      // one step of Bc independent sequences (policies.py:77-96 with S, M fed; act model nsteps = 1)
            const int ss = 2 * m->pi.nh;
            rc = lstm_forward(m->pi, in, params, ws.pi, Bc, 1, state_in ? state_in + (size_t)c0 * ss : nullptr, mask + c0,
                              nullptr, state_out + (size_t)c0 * ss, false, st);
            if (rc) return rc;
        }
        if (m->vf_copy) {
            rc = net_forward(m, m->vf, in, params, ws.vf, Bc, st);
            if (rc) return rc;
        }
        }
        HeadArgs a;
        fill_head_args(m, params, ws, a);
        if ((rc = pick_ts(a, false))) return rc;
        // act side of narrow latents (the MLP nets): a tile's samples are finished by one thread each, so 32-sample tiles leave a
        // 1024-env step on 32 workgroups walking serial expf / logf chains -- smaller tiles spread it over the chip (round 6:
        // heads_act 25 -> see profiles/README.md)
        if (a.nlat <= 128 && !wave_heads_shape(a)) {
            while (a.TS > 4 && (Bc + a.TS - 1) / a.TS < 2 * num_cus()) a.TS >>= 1;
        }
        a.Bc = Bc; a.row0 = c0;
        a.noise = noise; a.actions_out = actions_out; a.values_out = values_out; a.neglogp_out = neglogp_out;
        a.pdparam_out = pdparam_out;
        HeadLds L;
        size_t lds = head_lds_carve(a, false, nullptr, L);
        int ntiles = (Bc + a.TS - 1) / a.TS;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)heads_act_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        ProfScope ps("heads_act", 0.0, (double)Bc * (4.0 * a.nlat + (a.shared ? 0 : 4.0 * a.nlatv) + 16.0), st);
        const bool wave_ok = a.has_pi_head && a.shared && a.pd_kind == MRL_PD_CATEGORICAL && a.nact <= 8 && a.nlat == 512 &&
                             (uintptr_t)a.lat % 16 == 0 && get_option("heads_wave", "MRL_HEADS_WAVE", 2);
        if (wave_ok) {     // one wave per sample at a time; from ~1024 samples up a wave takes several (its 14 KB of head weights are
            // loaded once per wave with strided scalar loads: at num_envs = 4096 one sample per wave spent 32 us mostly on that)
            const int spw = std::max(1, std::min(8, Bc / 1024));
            hipLaunchKernelGGL(heads_act_wave_kernel<8>, dim3(std::max(1, std::min((Bc + 4 * spw - 1) / (4 * spw), 2048))), dim3(256), 0, st, a);
        }
        else
            hipLaunchKernelGGL(heads_act_kernel, dim3(std::min(ntiles, HEAD_MAXBLK)), dim3(256), lds, st, a);
        MRL_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int mrl_model_act(const mrl_model* m, const float* params, const void* obs, const float* noise, int n,
                             void* actions_out, float* values_out, float* neglogp_out, float* pdparam_out,
                             void* workspace, size_t workspace_bytes, int chunk, void* stream) {
    if (m && m->pi.lstm) return MRL_EINVAL;
    return model_act_impl(m, params, obs, noise, n, nullptr, nullptr, nullptr, actions_out, values_out, neglogp_out,
                          pdparam_out, workspace, workspace_bytes, chunk, stream);
}
extern "C" int mrl_model_act_rnn(const mrl_model* m, const float* params, const void* obs, const float* noise, int n,
                                 const float* state_in, const uint8_t* mask, float* state_out, void* actions_out,
                                 float* values_out, float* neglogp_out, float* pdparam_out, void* workspace,
                                 size_t workspace_bytes, int chunk, void* stream) {
    if (!m || !m->pi.lstm || !mask || !state_out) return MRL_EINVAL;
    return model_act_impl(m, params, obs, noise, n, state_in, mask, state_out, actions_out, values_out, neglogp_out,
                          pdparam_out, workspace, workspace_bytes, chunk, stream);
}

struct RnnIn { const float* states; const uint8_t* masks; int nseq; };    // recurrent inputs of a minibatch (model.py:153-155)

// Gradient of the loss over samples [mb0, mb0+mbn) of a minibatch of Bstat samples whose advantage statistics
// span the WHOLE minibatch: mb0 = 0, mbn = Bstat is Model.train (model.py:133-158); a proper sub-range is one
// MicrobatchedModel step (microbatched_model.py:40-60: normalise once, then per-slice losses with means over the slice).
static int model_grad_range(const mrl_model* m, const float* params, const void* obs, const void* actions,
                            const float* returns, const float* values, const float* neglogpacs,
                            const int64_t* idx, int Bstat, int mb0, int mbn, int T, int N, float cliprange, float ent_coef,
                            float vf_coef, float* grads_out, float* stats_out, void* workspace,
                            size_t workspace_bytes, int chunk, void* stream, bool want_sq = false, int* sqn_out = nullptr,
                            const RnnIn* rnn = nullptr) {
    if (sqn_out) *sqn_out = 0;
    if (m && m->pi.lstm) {
        // recurrent policies: whole minibatch of nseq trajectories of B/nseq steps each, sample b = seq*steps + t, one chunk
        if (!rnn || !rnn->masks || rnn->nseq <= 0 || mb0 != 0 || mbn != Bstat || Bstat % rnn->nseq != 0) return MRL_EINVAL;
        if (Bstat > chunk) return MRL_EUNSUP;
    } else if (rnn) return MRL_EINVAL;
    if (!m || !params || !obs || !actions || !returns || !values || !neglogpacs || !grads_out || !stats_out ||
        !workspace || Bstat <= 0 || chunk <= 0 || mb0 < 0 || mbn <= 0 || mb0 + mbn > Bstat)
        return MRL_EINVAL;
    if (idx && (T <= 0 || N <= 0)) return MRL_EINVAL;
    if (idx && (long)T * N > 0x7fffffffL) return MRL_EUNSUP;
    hipStream_t st = (hipStream_t)stream;
    const bool whole = mb0 == 0 && mbn == Bstat;
    const int B = mbn;
    Ws ws;
    carve(m, chunk, (char*)workspace, ws);
    if (ws.total > workspace_bytes) return MRL_ENOSPC;
    StepCtx ctx;
    ctx.comm = m->comm;
    ctx.rank_weight = m->rank_weight;
    if (ctx.comm && !m->vf_copy && m->pi.L.size() >= 2) {       // flat layout: pi layers in order, then the heads
        ctx.early_layer = (int)m->pi.L.size() - 1;
        ctx.early_lo = m->pi.L.back().w_off;
    }
    // global-norm partials ride on the slab reductions when the gradient is final after one pass and stays local
    if (want_sq && !ctx.comm && mbn <= chunk && get_option("fused_norm", "MRL_FUSED_NORM", 1)) {
        ctx.sqpart = ws.sqpart;
        ctx.sqcap = SQ_MAX_PART;
    }
    double* advpart = ws.dscratch;
    double* spart = ws.dscratch + ADV_G * 2;
    double* stats_acc = spart + SPART_MAX * 5;
    // fused whole-step kernel for the 2 x 64 tanh MLP (mlpstep.hip.h)?  It computes the advantage statistics itself.
    const int B0 = mbn;
    const bool mlp_fused = [&] {
        const mrl_model_desc& d = m->d;
        const int K0 = (int)m->ob_elems, nets = m->vf_copy ? 2 : 1, ntiles = (B0 + 31) / 32;
        return get_option("mlp_fused", "MRL_MLP_FUSED", 1) && d.network == MRL_NET_MLP && d.num_layers == 2 && !d.layer_norm && d.pd_kind <= MRL_PD_DIAG_GAUSSIAN &&
               d.num_hidden == MLP_NH && d.activation == MRL_ACT_TANH && m->has_pi_head && d.nact <= 32 && K0 % 4 == 0 &&
               B0 <= chunk && ntiles <= MLP_MAX_TILES && mlp_step_lds_bytes(K0, nets) <= 160 * 1024 &&
               (size_t)ntiles * m->P <= ws.part_floats && (uintptr_t)params % 16 == 0 && (uintptr_t)obs % 16 == 0;
    }();
    const float* stat_ret = returns; const float* stat_val = values; const int64_t* stat_idx = idx;
    // minibatch advantage statistics (model.py:136-139)
    if (!mlp_fused) {
        int G = std::min(ADV_G, (Bstat + 255) / 256);
        ProfScope* psadv = new ProfScope("adv_stats", 0.0, (idx ? 16.0 : 8.0) * Bstat, st);
        hipLaunchKernelGGL(advstat_part_kernel, dim3(G), dim3(256), 0, st, returns, values, idx, Bstat, T, N, advpart,
                           whole ? ws.srow : nullptr, chunk);
        MRL_LAUNCH_CHECK();
        hipLaunchKernelGGL(advstat_final_kernel, dim3(1), dim3(256), 0, st, advpart, G, Bstat, ws.advstat, stats_acc);
        delete psadv;
        MRL_LAUNCH_CHECK();
    }
    const size_t ob_bytes = (size_t)m->ob_elems * (m->d.ob_dtype == MRL_OB_U8 ? 1 : 4);
    if (!whole) {                                   // the slice starts at sample mb0 of the minibatch
        if (idx) {
            idx += mb0;
        } else {
            const size_t act_bytes = m->d.pd_kind == MRL_PD_CATEGORICAL ? 4 : m->d.pd_kind == MRL_PD_MULTICATEGORICAL ? 4 * (size_t)m->d.nsub : 4 * (size_t)m->d.nact;
            obs = (const char*)obs + (size_t)mb0 * ob_bytes;
            actions = (const char*)actions + (size_t)mb0 * act_bytes;
            returns += mb0; values += mb0; neglogpacs += mb0;
        }
    }
    const float invB = 1.f / (float)B;
    // ---- fused whole-step kernel for the 2 x 64 tanh MLP (mlpstep.hip.h): 3 launches per step
    //      (step kernel incl. advantage statistics and index translation; slab reduction incl. the global-norm partials
    //      and the loss statistics; [clip +] Adam in mrl_model_train_step)
    if (mlp_fused) {
        const mrl_model_desc& d = m->d;
        const int ntiles = (B + 31) / 32;
        const int K0 = (int)m->ob_elems;
        const int nets = m->vf_copy ? 2 : 1;
        MlpStepArgs a;
        memset(&a, 0, sizeof a);
        for (int n = 0; n < nets; ++n) {
            const Net& net = n == 0 ? m->pi : m->vf;
            a.w0[n] = net.L[0].w_off; a.b0[n] = net.L[0].b_off; a.w1[n] = net.L[1].w_off; a.b1[n] = net.L[1].b_off;
        }
        a.wpi = m->pi_w; a.bpi = m->pi_b; a.logstd = m->logstd; a.wvf = m->vf_w; a.bvf = m->vf_b;
        a.K0 = K0; a.nact = d.nact; a.nets = nets; a.pd_kind = d.pd_kind; a.P = m->P;
        a.params = params; a.obs = (const float*)obs; a.actions = actions; a.returns = returns; a.values = values;
        a.neglogp = neglogpacs; a.advstat = m->ext_advstat; a.cliprange = cliprange; a.ent_coef = ent_coef;
        a.vf_coef = vf_coef; a.invB = invB; a.B = B; a.part = ws.part; a.spart = spart;
        a.stat_ret = stat_ret; a.stat_val = stat_val; a.stat_idx = stat_idx; a.Bstat = Bstat; a.T = T; a.N = N;
        a.tile_idx = idx;                               // already advanced to the slice (nullptr: direct rows)
        a.srow = nullptr;
        {   // MRL_MLP_DBG=1: phase timestamps of workgroup 0 land in the last 64 bytes of the zero page
            const int dbgon = dbg_option("mlp_dbg", "MRL_MLP_DBG");
            a.dbg = dbgon ? reinterpret_cast<long long*>(ws.zeros) + 64 : nullptr;
            a.dbg_block = dbgon - 1;                    // MRL_MLP_DBG = 1 + the workgroup to stamp
        }
        const int waves8 = get_option("mlp_waves", "MRL_MLP_WAVES", 8) >= 8;      // 8 waves per workgroup (4: the round-2 form)
        // separate policy / value nets: one workgroup per (tile, net) -- half the serial chain, twice the workgroups
        a.slice = (nets == 2 && get_option("mlp_slice", "MRL_MLP_SLICE", 1)) ? 1 : 0;
        const int nwg = ntiles * (a.slice ? 2 : 1);
        const size_t lds = mlp_step_lds_bytes(K0, a.slice ? 1 : nets);
        MRL_HIP_CHECK(raise_lds_limit((const void*)mlp_step_kernel<512>));         // once per (device, kernel)
        MRL_HIP_CHECK(raise_lds_limit((const void*)mlp_step_kernel<256>));
        {
            // algorithmic flops: fwd + bwd of both nets on B samples
            double fl = 0.0;
            for (int n = 0; n < nets; ++n) fl += 2.0 * B * ((double)K0 * 64 * 2 + 64.0 * 64 * 3);
            ProfScope ps("mlp_step", fl, 0.0, st);
            if (waves8) hipLaunchKernelGGL(mlp_step_kernel<512>, dim3(nwg), dim3(512), lds, st, a);
            else hipLaunchKernelGGL(mlp_step_kernel<256>, dim3(nwg), dim3(256), lds, st, a);
        }
        MRL_LAUNCH_CHECK();
        int rc = reduce_slabs(ws.part, m->P, ntiles, grads_out, m->P, 0, st, &ctx, spart, ntiles, invB, stats_out);
        if (rc) return rc;
        if (ctx.comm) {
            if ((rc = comm_allreduce_async(ctx.comm, grads_out, m->P, ctx.rank_weight, st))) return rc;
            if ((rc = comm_join(ctx.comm, st))) return rc;
        }
        if (sqn_out) *sqn_out = ctx.sqn;
        return 0;
    }
    for (int c0 = 0; c0 < B; c0 += chunk) {
        const int Bc = std::min(chunk, B - c0);
        const int accumulate = c0 > 0;
        ctx.final_chunk = c0 + chunk >= B;
        In in;
        if (idx) {
            if (c0 > 0 || !whole) {                      // chunk 0 of a whole minibatch was translated by advstat_part_kernel
                hipLaunchKernelGGL(translate_idx_kernel, dim3((Bc + 255) / 256), dim3(256), 0, st, idx + c0, Bc, T, N, ws.srow);
                MRL_LAUNCH_CHECK();
            }
            in.obs = obs; in.srow = ws.srow;
        } else {
            in.obs = (const char*)obs + (size_t)c0 * ob_bytes; in.srow = nullptr;
        }
        // (ws.part: split-K of an fc layer whose tiles would not fill the chip -- fc1 at a minibatch of 8192 is 256 tiles of 128 x 128 for
        // 512 slots, each walking K = 3136 alone: 186 us where 131072 rows take 119 us per 8192; the backward pass's slabs come later on
        // the same stream)
        // Only when the whole range is that small: the remainder chunk of a large minibatch keeps the unsplit sums, so that the
        // result of a large minibatch does not depend on how it is chunked beyond the order of the per-chunk weight-gradient sums
        // (bench.py's self-check compares two chunkings of one 131072-sample minibatch).
        const bool small_range = B <= 16384;
        int rc = net_forward(m, m->pi, in, params, ws.pi, Bc, st, small_range ? ws.part : nullptr, small_range ? ws.part_floats : 0);
        if (rc) return rc;
        if (m->pi.lstm && (rc = lstm_forward(m->pi, in, params, ws.pi, rnn->nseq, Bc / rnn->nseq, rnn->states, rnn->masks,
                                             in.srow, nullptr, true, st)))
            return rc;
        if (m->vf_copy && (rc = net_forward(m, m->vf, in, params, ws.vf, Bc, st))) return rc;
        HeadArgs a;
        fill_head_args(m, params, ws, a);
        if ((rc = pick_ts(a, true))) return rc;
        a.Bc = Bc; a.row0 = c0;
        a.actions = actions; a.returns = returns; a.values = values; a.neglogp = neglogpacs;
        a.idx = idx ? idx + c0 : nullptr; a.T = T; a.N = N;
        a.advstat = ws.advstat; a.cliprange = cliprange; a.ent_coef = ent_coef; a.vf_coef = vf_coef; a.invB = invB;
        a.dz_pi = ws.pi.dlat();
        a.dz_vf = m->vf_copy ? ws.vf.dz.back() : nullptr;
        a.hpart = ws.part; a.spart = spart;
        HeadLds L;
        size_t lds = head_lds_carve(a, true, nullptr, L);
        int ntiles = (Bc + a.TS - 1) / a.TS;
        int nblk = std::min(ntiles, HEAD_MAXBLK);
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)heads_train_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        // wave-per-sample kernel for the NatureCNN head shape (Categorical <= 8 actions, shared 512-wide latent)
        const bool wave_ok = a.has_pi_head && a.shared && a.pd_kind == MRL_PD_CATEGORICAL && a.nact <= 8 && a.nlat == 512 &&
                             m->HP == a.nlat * a.nact + a.nact + a.nlat + 1 && (uintptr_t)a.lat % 16 == 0 &&
                             (uintptr_t)a.dz_pi % 16 == 0 && get_option("heads_wave", "MRL_HEADS_WAVE", 2);
        if (wave_ok) nblk = std::min((Bc + 31) / 32, HEAD_MAXBLK);       // >= 8 samples per wave
        {
            // algorithmic traffic: latent in, dz out, per-sample rollout scalars
            ProfScope ps("heads_loss", 0.0, (double)Bc * (8.0 * a.nlat + (a.shared ? 0 : 8.0 * a.nlatv) + 28.0), st);
            if (wave_ok) {
                const size_t wl = (size_t)4 * m->HP * sizeof(float);
                if (wl > 64 * 1024) MRL_HIP_CHECK(raise_lds_limit((const void*)heads_train_wave_kernel<8>));        // once per (device, kernel)
                if (a.nact <= 6 && get_option("heads_wave", "MRL_HEADS_WAVE", 2) >= 2) {
                    if (wl > 64 * 1024) MRL_HIP_CHECK(raise_lds_limit((const void*)heads_train_wave2_kernel<8, 6>));
                    hipLaunchKernelGGL((heads_train_wave2_kernel<8, 6>), dim3(nblk), dim3(256), wl, st, a);
                } else {
                    hipLaunchKernelGGL(heads_train_wave_kernel<8>, dim3(nblk), dim3(256), wl, st, a);
                }
            } else {
                hipLaunchKernelGGL(heads_train_kernel, dim3(nblk), dim3(256), lds, st, a);
            }
        }
        MRL_LAUNCH_CHECK();
        if ((rc = reduce_slabs(ws.part, m->HP, nblk, grads_out + m->head_off, m->HP, accumulate, st, &ctx))) return rc;
        hipLaunchKernelGGL(heads_stats_reduce_kernel, dim3(1), dim3(256), 0, st, spart, nblk, stats_acc);
        MRL_LAUNCH_CHECK();
        if (m->pi.lstm && (rc = lstm_backward(m->pi, in, params, ws.pi, ws, grads_out, rnn->nseq, Bc / rnn->nseq, rnn->masks,
                                              in.srow, st, ctx)))
            return rc;
        if ((rc = net_backward<kExp>(m, m->pi, in, params, ws.pi, ws, grads_out, Bc, accumulate, st, ctx, true))) return rc;
        if (m->vf_copy && (rc = net_backward<kExp>(m, m->vf, in, params, ws.vf, ws, grads_out, Bc, accumulate, st, ctx, false)))
            return rc;
    }
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(64), 0, st, stats_acc, invB, stats_out);
    MRL_LAUNCH_CHECK();
    if (ctx.comm) {        // the rest of the flat gradient, then the compute stream waits for both collectives
        int rc = comm_allreduce_async(ctx.comm, grads_out, ctx.early_layer >= 0 ? ctx.early_lo : m->P, ctx.rank_weight, st);
        if (rc) return rc;
        if ((rc = comm_join(ctx.comm, st))) return rc;
    }
    if (sqn_out) *sqn_out = ctx.sqn;
    return 0;
}

extern "C" int mrl_model_grad(const mrl_model* m, const float* params, const void* obs, const void* actions,
                              const float* returns, const float* values, const float* neglogpacs,
                              const int64_t* idx, int B, int T, int N, float cliprange, float ent_coef,
                              float vf_coef, float* grads_out, float* stats_out, void* workspace,
                              size_t workspace_bytes, int chunk, void* stream) {
    return model_grad_range(m, params, obs, actions, returns, values, neglogpacs, idx, B, 0, B, T, N, cliprange, ent_coef,
                            vf_coef, grads_out, stats_out, workspace, workspace_bytes, chunk, stream);
}

extern "C" int mrl_model_grad_micro(const mrl_model* m, const float* params, const void* obs, const void* actions,
                                    const float* returns, const float* values, const float* neglogpacs,
                                    const int64_t* idx, int B, int mb0, int mbn, int T, int N, float cliprange,
                                    float ent_coef, float vf_coef, float* grads_out, float* stats_out, void* workspace,
                                    size_t workspace_bytes, int chunk, void* stream) {
    return model_grad_range(m, params, obs, actions, returns, values, neglogpacs, idx, B, mb0, mbn, T, N, cliprange,
                            ent_coef, vf_coef, grads_out, stats_out, workspace, workspace_bytes, chunk, stream);
}

// ---- data-parallel attachment --- common/mpi_adam_optimizer.py:18-51 -----------------------------------------------
extern "C" int mrl_advstat_minibatches(const float* returns, const float* values, const int64_t* idx, int nmb, int B, int T,
                                       int N, float* out, void* stream) {
    if (!returns || !values || !idx || !out || nmb <= 0 || B <= 0 || T <= 0 || N <= 0) return MRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // the thread count of the fused step kernel in effect: the partial sums must be the ones its workgroups would form
    if (get_option("mlp_waves", "MRL_MLP_WAVES", 8) >= 8)
        hipLaunchKernelGGL(mlp_advstat_kernel<512>, dim3(nmb), dim3(512), 0, st, returns, values, idx, B, T, N, out);
    else
        hipLaunchKernelGGL(mlp_advstat_kernel<256>, dim3(nmb), dim3(256), 0, st, returns, values, idx, B, T, N, out);
    MRL_LAUNCH_CHECK();
    return 0;
}
extern "C" int mrl_model_set_advstat(mrl_model* m, const float* advstat) {
    if (!m) return MRL_EINVAL;
    m->ext_advstat = advstat;
    return 0;
}

extern "C" int mrl_model_attach_comm(mrl_model* m, mrl_comm* comm, float rank_weight) {
    if (!m || rank_weight <= 0.f) return MRL_EINVAL;
    m->comm = comm;
    m->rank_weight = rank_weight;
    return 0;
}

// ---- Model.train in ONE call --- ppo2/model.py:133-158 + :97-114 ---------------------------------------------------
// gather -> forward -> loss -> backward -> [RCCL all-reduce] -> / total weight -> clip_by_global_norm -> Adam.
// Single-rank, single-chunk steps take the global norm from the partials the gradient's slab reductions leave behind.
extern "C" int mrl_model_train_step(const mrl_model* m, float* params, float* grads, float* adam_m, float* adam_v,
                                    const void* obs, const void* actions, const float* returns, const float* values,
                                    const float* neglogpacs, const int64_t* idx, int B, int T, int N, float cliprange,
                                    float ent_coef, float vf_coef, float alpha, const float* alpha_dev, float beta1,
                                    float beta2, float eps, float max_grad_norm, float total_weight, float* stats_out,
                                    float* gnorm_out, void* workspace, size_t workspace_bytes, int chunk, void* stream) {
    if (!m || !adam_m || !adam_v || total_weight <= 0.f) return MRL_EINVAL;
    int sqn = 0;
    const bool want_sq = max_grad_norm >= 0.f && total_weight == 1.f;
    int rc = model_grad_range(m, params, obs, actions, returns, values, neglogpacs, idx, B, 0, B, T, N, cliprange, ent_coef,
                              vf_coef, grads, stats_out, workspace, workspace_bytes, chunk, stream, want_sq, &sqn);
    if (rc) return rc;
    Ws ws;
    carve(m, chunk, (char*)workspace, ws);
    return adam_clip_apply(params, grads, adam_m, adam_v, m->P, alpha, alpha_dev, beta1, beta2, eps, max_grad_norm,
                           total_weight, gnorm_out, ws.sqpart, sqn > 0 ? ws.sqpart : nullptr, sqn, (hipStream_t)stream);
}

// ---- recurrent policies --- ppo2/model.py:153-155, ppo2/ppo2.py:167-180 -----------------------------
extern "C" int mrl_model_grad_rnn(const mrl_model* m, const float* params, const void* obs, const void* actions,
                                  const float* returns, const float* values, const float* neglogpacs,
                                  const uint8_t* masks, const float* states, int nseq, const int64_t* idx, int B, int T,
                                  int N, float cliprange, float ent_coef, float vf_coef, float* grads_out, float* stats_out,
                                  void* workspace, size_t workspace_bytes, int chunk, void* stream) {
    if (!m || !m->pi.lstm) return MRL_EINVAL;
    RnnIn r{states, masks, nseq};
    return model_grad_range(m, params, obs, actions, returns, values, neglogpacs, idx, B, 0, B, T, N, cliprange, ent_coef,
                            vf_coef, grads_out, stats_out, workspace, workspace_bytes, chunk, stream, false, nullptr, &r);
}

#include "qnet.hip.h"
