/*
 * mrl.h -- C ABI of libmrl.so, the MI355X (gfx950) PPO2 / DQN-replay hot path.
 *
 * The reference (openai/baselines) has no FFI: its extension points are Python call
 * signatures.  This header is the boundary a non-Python host (or the Python
 * mirror in baselines_amd/) binds instead of `sess.run(...)` / the NumPy loops.  Each entry
 * point cites the reference code it replaces (paths relative to baselines/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous on it;
 *   - return value: 0 ok, >0 hipError_t, <0 argument error (MRL_E*);
 *   - no hidden allocation on the data path: scratch comes from a caller-provided workspace whose
 *     size is queried first.  Process-global state is limited to three opt-in tables, all documented
 *     at the end of this header: the HIP-event profiler (mrl_prof_*), the engine options
 *     (mrl_set_option / MRL_* environment variables, read once) and the tile-tuning overrides
 *     (mrl_tune_set).  Calls on ONE mrl_model / mrl_comm must not race; different objects may be used
 *     from different threads;
 *   - rollout storage is time-major SoA [T][N]; the reference's env-major flat index
 *     i = e*T + t (ppo2/runner.py:69-74 `sf01`) is translated inside the kernels, so sf01's
 *     full copy never happens.
 */
#ifndef MRL_H_
#define MRL_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRL_VERSION 1

#define MRL_EINVAL   (-1)   /* bad argument / unsupported shape */
#define MRL_ENOSPC   (-2)   /* workspace too small */
#define MRL_EUNSUP   (-3)   /* configuration outside the supported hot path */
#define MRL_ECOMM    (-4)   /* RCCL error / librccl unavailable (text: mrl_comm_last_error()) */

int mrl_version(void);
const char* mrl_strerror(int code);
/* number of visible HIP devices (<=0: none).  Used by the Python loader to fail loudly. */
int mrl_device_count(void);

/* ---- K2: GAE(lambda) backward recurrence --- ppo2/runner.py:52-65 -----------------------
 * rew,val f32 [T][N]; done u8 [T][N] (flag ENTERING step t); last_val f32 [N]; last_done u8 [N].
 * Reproduces NumPy's dtype mix exactly: f32 product gamma*V_{t+1}, f64 carry, RN store to f32;
 * ret = f32(adv_f32 + val).  adv_out may be NULL.  Bit-exact vs the reference. */
int mrl_gae(const float* rew, const float* val, const uint8_t* done, const float* last_val,
            const uint8_t* last_done, double gamma, double lam, float* adv_out, float* ret_out,
            int T, int N, void* stream);

/* ---- K4: standalone minibatch gather --- ppo2/ppo2.py:162-164 `arr[mbinds]` ---------------
 * src time-major rows [T*N][row_bytes]; idx env-major flat indices int64 [B]; dst [B][row_bytes]. */
int mrl_gather_rows(const void* src, const int64_t* idx, void* dst, int B, int T, int N,
                    int row_bytes, void* stream);
/* sf01 as a device op (only for callers that insist on env-major copies): dst[e*T+t] = src[t*N+e] */
int mrl_sf01(const void* src, void* dst, int T, int N, int row_bytes, void* stream);

/* ---- model description --- common/policies.py:121-179, common/models.py:15-26,74-103 ------ */
enum { MRL_NET_MLP = 0, MRL_NET_NATURE_CNN = 1, MRL_NET_LSTM = 2, MRL_NET_CNN_LSTM = 3,
       MRL_NET_CONV_ONLY = 4 /* Q-networks only */ };                            /* models.py:74-249 */
enum { MRL_PD_CATEGORICAL = 0, MRL_PD_DIAG_GAUSSIAN = 1,
       MRL_PD_MULTICATEGORICAL = 2,   /* MultiDiscrete(nvec): independent categoricals over the slices of the flat logits (distributions.py:206-225) */
       MRL_PD_BERNOULLI = 3 };         /* MultiBinary(n): independent Bernoullis, logits -> sigmoid (distributions.py:253-276) */
enum { MRL_OB_F32 = 0, MRL_OB_U8 = 1 };
enum { MRL_ACT_NONE = 0, MRL_ACT_RELU = 1, MRL_ACT_TANH = 2 };

typedef struct mrl_model_desc {
    int network;          /* MRL_NET_* */
    int ob_ndim;          /* 1 (vector) or 3 (H,W,C image) */
    int ob_shape[3];
    int ob_dtype;         /* MRL_OB_* */
    int num_layers;       /* mlp: models.py:75 (default 2) */
    int num_hidden;       /* mlp: (default 64) */
    int activation;       /* mlp: MRL_ACT_TANH (default) | MRL_ACT_RELU */
    int value_copy;       /* policies.py:154-166: 0 = shared latent, 1 = value_network='copy' */
    int pd_kind;          /* MRL_PD_*  (distributions.py:278-290) */
    int nact;             /* Discrete.n or Box.shape[0] */
    int nlstm;            /* lstm / cnn_lstm: hidden state size (models.py:132, default 128).  32 / 64 / 96 / 128: recurrent weights
                           * register-resident for the whole scan; any other width up to 1024 (e.g. impala_cnn_lstm's 256): streamed from L2 */
    int layer_norm;       /* mlp(layer_norm=True), models.py:97-98: tf.contrib.layers.layer_norm(h, center=True, scale=True)
                           * between fc and activation; variables <scope>/LayerNorm[_i]/{beta,gamma} after each layer's w, b */
    /* conv stack of MRL_NET_NATURE_CNN / MRL_NET_CNN_LSTM (models.py:15-26, 106-129).  nconv = 0: nature_cnn's
     * (32, 8, 4), (64, 4, 2), (64, 3, 1) + fc 512.  nconv in 1..4: layers c1..c<nconv> with (filters, kernel size, stride),
     * filters % 4 == 0, then conv_to_fc and fc1 of fc_hidden units -- cnn_small is nconv = 2, (8, 8, 4), (16, 4, 2), fc_hidden
     * = 128.  conv_pad: 0 = 'VALID' (the default of a2c/utils.py:37 conv), 1 = 'SAME' (cnn(pad='SAME'), a conv_kwargs entry). */
    int nconv;  int convs[4][3];  int fc_hidden;  int conv_pad;
    /* MRL_PD_MULTICATEGORICAL: nsub categoricals with nvec[i] classes each, nact = sum nvec (the width of the flat pdparam);
     * actions are int32 [n][nsub].  MRL_PD_BERNOULLI: nact bits, actions int32 [n][nact] (0 / 1); both sample from
     * uniform(0, 1) noise [n][nact] (Gumbel-max per slice / u < sigmoid(logit)). */
    int nsub;  int nvec[16];
} mrl_model_desc;

typedef struct mrl_model mrl_model;   /* host-side layout object; owns no device memory */

int  mrl_model_create(const mrl_model_desc* desc, mrl_model** out);
void mrl_model_destroy(mrl_model* m);
long mrl_model_num_params(const mrl_model* m);          /* P (floats) */
int  mrl_model_num_tensors(const mrl_model* m);
/* tensor i in TF variable-creation order: name (e.g. "ppo2_model/pi/c1/w"),
 * shape (<=4 dims, TF shapes incl. conv bias [1,nf,1,1]), flat offset, orthogonal-init scale as the
 * f64 the reference multiplies with (np.sqrt(2), 0.01, 1.0; <0: zeros). */
int  mrl_model_tensor_info(const mrl_model* m, int i, char* name, int name_cap, int* ndim,
                           int shape[4], long* offset, double* init_scale);
/* workspace bytes needed to process `chunk` samples at a time (forward+backward).  The caller zero-fills the
 * workspace ONCE after allocating it (it contains a page of zeros that no kernel writes); afterwards it is scratch. */
size_t mrl_model_workspace_bytes(const mrl_model* m, int chunk);

/* ---- K6+K10: act side --- common/policies.py:77-113, distributions.py:199-201,247-248 -----
 * obs [n][ob...] in the model's ob dtype.  noise: uniform(0,1) f32 [n][nact] (Categorical,
 * Gumbel-max) or N(0,1) f32 [n][nact] (DiagGaussian).  actions_out: int32 [n] or f32 [n][nact].
 * Any of actions_out/neglogp_out may be NULL together (value-only call == PolicyWithValue.value). */
int mrl_model_act(const mrl_model* m, const float* params, const void* obs, const float* noise,
                  int n, void* actions_out, float* values_out, float* neglogp_out,
                  float* pdparam_out /* [n][nact] logits or mean, may be NULL */,
                  void* workspace, size_t workspace_bytes, int chunk, void* stream);

/* ---- recurrent policies --- common/models.py:132-210 (`lstm`, `cnn_lstm`), a2c/utils.py:81-102 ----------
 * State of one environment: f32 [2*nlstm] = (c | h) (utils.py:87,101); mrl_model_state_size = 2*nlstm (0: not recurrent).
 * act: one step of n independent sequences -- state_in (NULL: zeros) and mask u8 [n] (done flag entering the step: the
 * state is zeroed where it is set) in, state_out out; otherwise as mrl_model_act (which rejects recurrent models).
 * grad: a minibatch of nseq WHOLE trajectories (ppo2/ppo2.py:167-180): B = nseq * steps samples in env-major order
 * b = seq*steps + t, given directly (idx == NULL) or as env-major flat indices into the time-major rollout (then masks
 * is the rollout's done array like the other fields); states f32 [nseq][2*nlstm] = state of each sequence BEFORE its
 * first step (model.py:153-155: S = states, M = masks); back-propagation through all steps of the minibatch. */
int mrl_model_state_size(const mrl_model* m);
int mrl_model_act_rnn(const mrl_model* m, const float* params, const void* obs, const float* noise, int n,
                      const float* state_in, const uint8_t* mask, float* state_out, void* actions_out,
                      float* values_out, float* neglogp_out, float* pdparam_out, void* workspace,
                      size_t workspace_bytes, int chunk, void* stream);
int mrl_model_grad_rnn(const mrl_model* m, const float* params, const void* obs, const void* actions,
                       const float* returns, const float* values, const float* neglogpacs, const uint8_t* masks,
                       const float* states, int nseq, const int64_t* idx, int B, int T, int N, float cliprange,
                       float ent_coef, float vf_coef, float* grads_out, float* stats_out, void* workspace,
                       size_t workspace_bytes, int chunk, void* stream);

/* ---- K4+K5+K6+K7: gradient of the PPO2 loss on one minibatch --- ppo2/model.py:57-91,133-158
 * Rollout fields are time-major [T*N] (obs [T*N][ob...]); idx int64 [B] are the reference's
 * env-major flat indices (idx == NULL: rows 0..B-1 of the given arrays, T and N ignored).
 * actions: int32 [T*N] (Categorical) or f32 [T*N][nact].
 * Computes advs = returns - values normalised over the MINIBATCH (model.py:136-139), the loss,
 * closed-form gradients and back-propagates through heads and networks.
 * grads_out f32 [P] is overwritten with dloss/dparams (NOT yet clipped).
 * stats_out f32 [5] = [pg_loss, vf_loss, entropy, approxkl, clipfrac] (model.py:115-116). */
int mrl_model_grad(const mrl_model* m, const float* params, const void* obs, const void* actions,
                   const float* returns, const float* values, const float* neglogpacs,
                   const int64_t* idx, int B, int T, int N, float cliprange, float ent_coef,
                   float vf_coef, float* grads_out, float* stats_out, void* workspace,
                   size_t workspace_bytes, int chunk, void* stream);

/* ---- MicrobatchedModel step --- ppo2/microbatched_model.py:36-60 ----------
 * As mrl_model_grad, but the advantage statistics span all B samples of the minibatch while the
 * loss / gradient / stats cover only samples [mb0, mb0+mbn) (means over mbn, like the reference's
 * microbatch-sized graph). */
int mrl_model_grad_micro(const mrl_model* m, const float* params, const void* obs, const void* actions,
                         const float* returns, const float* values, const float* neglogpacs,
                         const int64_t* idx, int B, int mb0, int mbn, int T, int N, float cliprange,
                         float ent_coef, float vf_coef, float* grads_out, float* stats_out,
                         void* workspace, size_t workspace_bytes, int chunk, void* stream);
/* acc := (first ? 0 : acc) + clip_by_global_norm(grads / total_weight): the reference sums its
 * post-clip `self.grads` per microbatch (microbatched_model.py:57-64, model.py:105-112); the
 * average and the un-clipped Adam apply are mrl_adam_clip_step(acc, max_grad_norm < 0,
 * total_weight = nmicrobatches).  scratch: >= mrl_adam_scratch_bytes(P). */
int mrl_clip_accumulate(const float* grads, float* acc, long P, float max_grad_norm, float total_weight,
                        int first, void* scratch, void* stream);

/* ---- K8+K9: [rank average] -> clip_by_global_norm -> Adam --- ppo2/model.py:97-114,
 * common/mpi_adam_optimizer.py:39-40.  grads := grads / total_weight (if != 1) BEFORE the norm,
 * scale = c*min(1/gn, 1/c) (max_grad_norm < 0: no clipping), TF-1 ApplyAdam update with
 * alpha = lr*sqrt(1-beta2_power)/(1-beta1_power) supplied by the caller (host f32).
 * scratch: >= mrl_adam_scratch_bytes(P).  gnorm_out f32 [1] may be NULL. */
size_t mrl_adam_scratch_bytes(long P);
int mrl_adam_clip_step(float* params, float* grads, float* m, float* v, long P, float alpha,
                       float beta1, float beta2, float eps, float max_grad_norm,
                       float total_weight, float* gnorm_out, void* scratch, void* stream);
/* Same, with the step size read from device memory (f32 [1]) when the kernel runs: the launch carries no
 * per-step scalar, so a captured hipGraph of an epoch's minibatch steps can be replayed (the caller refreshes the
 * alpha array between replays -- TF keeps beta1_power/beta2_power in device variables for the same reason). */
int mrl_adam_clip_step_dev(float* params, float* grads, float* m, float* v, long P, const float* alpha_dev,
                           float beta1, float beta2, float eps, float max_grad_norm,
                           float total_weight, float* gnorm_out, void* scratch, void* stream);

/* ---- Model.train in one call --- ppo2/model.py:133-158 (advs, sess.run of stats + _train_op) with
 * :97-114 (gradients -> [MpiAdam average] -> clip_by_global_norm -> Adam apply) ------------------
 * = mrl_model_grad followed by mrl_adam_clip_step on the same stream, with the Adam update fed by the
 * gradient kernels directly: on a single rank the global-norm partial sums are produced by the
 * reductions that write the gradient (no second pass over it).  `grads` [P] is left holding the
 * averaged, clipped gradient (`self.grads`, model.py:112).  alpha_dev != NULL: step size read from
 * device memory (f32 [1]) so the call can sit in a captured hipGraph; else `alpha` (host f32,
 * lr*sqrt(1-beta2^t)/(1-beta1^t)).  With a communicator attached (below) the gradient is
 * all-reduced inside (overlapped with the backward pass); pass total_weight = sum of rank weights. */
int mrl_model_train_step(const mrl_model* m, float* params, float* grads, float* adam_m, float* adam_v,
                         const void* obs, const void* actions, const float* returns, const float* values,
                         const float* neglogpacs, const int64_t* idx, int B, int T, int N, float cliprange,
                         float ent_coef, float vf_coef, float alpha, const float* alpha_dev, float beta1,
                         float beta2, float eps, float max_grad_norm, float total_weight, float* stats_out,
                         float* gnorm_out, void* workspace, size_t workspace_bytes, int chunk, void* stream);

/* ---- minibatch advantage statistics ahead of time --- ppo2/model.py:136-139 for every minibatch of an epoch at once
 * (ppo2.py:157-166: the epoch's permutation is known before its first step).  idx int64 [nmb][B] env-major indices;
 * out f32 [nmb][2] = (mean, std) of returns - values over each minibatch, formed exactly like the fused MLP step kernel
 * forms them in every workgroup (same partial sums, same order: bit-identical).
 * mrl_model_set_advstat(m, p): while p != NULL the fused MLP step (option "mlp_fused") reads its minibatch statistics
 * from p[0..1] at launch time instead of gathering the whole minibatch again in each of its workgroups; the other engines
 * ignore it.  The pointer is read when a gradient / train-step call is issued (or captured into a hipGraph); NULL restores
 * the default. */
int mrl_advstat_minibatches(const float* returns, const float* values, const int64_t* idx, int nmb, int B, int T, int N,
                            float* out, void* stream);
int mrl_model_set_advstat(mrl_model* m, const float* advstat);

/* ---- C1: data-parallel collectives over RCCL / xGMI --- common/mpi_adam_optimizer.py:18-51
 * (`flat_grad * mpi_rank_weight` -> Allreduce(SUM) -> / total weight), common/mpi_util.py:15-26
 * (sync_from_root).  One communicator per process (= per GPU).  Bootstrap like ncclCommInitRank: rank 0
 * obtains a unique id (host bytes), the host program ships it to the other ranks by any means (a file,
 * a socket, torch.distributed's store), every rank then calls mrl_comm_create with its current HIP device
 * set.  librccl is bound lazily (dlopen) on the first of these calls. */
#define MRL_COMM_ID_BYTES 128
typedef struct mrl_comm mrl_comm;
int  mrl_comm_unique_id(void* id_out /* host, MRL_COMM_ID_BYTES */);
int  mrl_comm_create(const void* id /* host */, int nranks, int rank, mrl_comm** out);
void mrl_comm_destroy(mrl_comm* c);
int  mrl_comm_size(const mrl_comm* c);
int  mrl_comm_rank(const mrl_comm* c);
const char* mrl_comm_last_error(void);
/* in-place sum of the flat gradient f32 [P] over all ranks, asynchronous on `stream` (mpi_adam_optimizer.py:39) */
int mrl_allreduce_grads(mrl_comm* c, float* grads, long P, void* stream);
/* in-place broadcast of nbytes from `root` (parameters, Adam slots; mpi_util.py:15-26), asynchronous on `stream` */
int mrl_broadcast_state(mrl_comm* c, void* buf, size_t nbytes, int root, void* stream);
/* Attach (comm != NULL) / detach (NULL) a communicator: afterwards mrl_model_grad, mrl_model_grad_micro and
 * mrl_model_train_step return the rank-weighted SUM over ranks of the gradient.  The all-reduce is issued from
 * inside the backward pass on the communicator's own stream -- the tail of the flat gradient (fc1 + heads, 95 % of
 * NatureCNN's parameters) as soon as fc1's weight gradient is complete, the rest at the end -- and the compute
 * stream is made to wait for it before the call returns (stream order; no host synchronisation).  The division by
 * the total weight happens in the optimizer step (`total_weight`), before the global norm, like
 * mpi_adam_optimizer.py:40.  Loss statistics stay rank-local like the reference's. */
int mrl_model_attach_comm(mrl_model* m, mrl_comm* comm, float rank_weight);

/* ---- K13: HBM replay ring --- deepq/replay_buffer.py:24-43 ----------------------------------
 * SoA ring buffers [maxsize][...]: obs_t / obs_tp1 raw bytes (ob_bytes per transition), act int32,
 * rew f32, done f32.  insert: rows (next_idx + j) % maxsize, j < n, from contiguous batches.
 * gather: rows idx[b] (int32) into contiguous outputs (`_encode_sample`). */
int mrl_replay_insert(void* obs_t_buf, void* obs_tp1_buf, int32_t* act_buf, float* rew_buf, float* done_buf,
                      long maxsize, long next_idx, int n, int ob_bytes, const void* obs_t,
                      const void* obs_tp1, const int32_t* act, const float* rew, const float* done,
                      void* stream);
int mrl_replay_gather(const void* obs_t_buf, const void* obs_tp1_buf, const int32_t* act_buf,
                      const float* rew_buf, const float* done_buf, const int32_t* idx, int B, int ob_bytes,
                      void* obs_t_out, void* obs_tp1_out, int32_t* act_out, float* rew_out, float* done_out,
                      void* stream);

/* ---- K14: sum / min segment trees --- common/segment_tree.py:4-145 --------------------------
 * f64 arrays of 2*capacity nodes, heap layout (node 1 = root, leaves at [capacity, 2*capacity)),
 * capacity a power of two; neutral elements 0.0 / +inf.  All updates have the reference's SEQUENTIAL
 * semantics (with duplicate leaves the last entry wins) and every ancestor is op(left, right).
 *   mrl_segtree_set       leaf values given (host computes priority**alpha with Python-float pow
 *                         for bit parity: deepq/replay_buffer.py:100-105, 169-191)
 *   mrl_segtree_set_ring  n consecutive ring slots from `start` (mod maxsize) all get `leaf`
 *                         (PrioritizedReplayBuffer.add of a batch: max_priority**alpha)
 *   mrl_per_update_from_td  device fast path: leaf = pow(|td|+eps, alpha) (ocml pow, <= 2 ulp off
 *                         libm) and *max_priority = max(*max_priority, |td|+eps)  (deepq.py:302)
 *   mrl_segtree_set_ring_dev  mrl_segtree_set_ring with leaf = pow(*max_priority, alpha) read from the device
 *                         word mrl_per_update_from_td maintains (no host read-back per `add`) */
int mrl_segtree_init(double* sum_tree, double* min_tree, long capacity, void* stream);
int mrl_segtree_set(double* sum_tree, double* min_tree, long capacity, const int32_t* idx,
                    const double* leaf, int n, void* stream);
int mrl_segtree_set_ring(double* sum_tree, double* min_tree, long capacity, long start, long maxsize, int n,
                         double leaf, void* stream);
int mrl_per_update_from_td(double* sum_tree, double* min_tree, long capacity, const int32_t* idx,
                           const float* td, double eps, double alpha, double* max_priority, int n,
                           void* stream);
int mrl_segtree_set_ring_dev(double* sum_tree, double* min_tree, long capacity, long start, long maxsize, int n,
                             const double* max_priority, double alpha, void* stream);
/* stratified proportional sampling + importance weights --- deepq/replay_buffer.py:107-115, 155-165
 * uniforms f64 [B] are the `random.random()` draws; p_total = sum(0, length-1) EXCLUDES the newest
 * element (reference quirk, segment_tree.py:69-74); idx_out int32 [B] bit-exact vs the reference;
 * weights_out f64 [B] and/or weights_f32_out may be NULL.  length >= 2 (the reference recurses
 * forever on 1). */
int mrl_per_sample(const double* sum_tree, const double* min_tree, long capacity, long length, int B,
                   const double* uniforms, double beta, int32_t* idx_out, double* weights_out,
                   float* weights_f32_out, void* stream);

/* ---- K15: double-Q TD target + Huber --- deepq/build_graph.py:396-413, common/tf_util.py:39-45
 * q_* f32 [B][nA]; q_tp1_online == NULL: plain max over the target net.  td_out f32 [B];
 * loss_out f32 [1] = mean(w * huber(td)); dq_out (may be NULL) f32 [B][nA] = d loss / d q_t. */
size_t mrl_dqn_td_scratch_bytes(int B);
int mrl_dqn_td(const float* q_t, const float* q_tp1_target, const float* q_tp1_online, const int32_t* act,
               const float* rew, const float* done, const float* weights, float gamma, int B, int nA,
               float* td_out, float* loss_out, float* dq_out, void* scratch, void* stream);

/* ---- DQN Q-network --- deepq/models.py:5-45 (build_q_func), deepq/build_graph.py:146-199, 380-444 ----------
 * q_func = network(X) -> flatten -> action_value [fc(h) relu]* fc(nact) [+ dueling: state_value [fc(h) relu]* fc(1),
 * q = V + A - mean_a A].  network: MRL_NET_MLP / MRL_NET_NATURE_CNN (common/models.py, orthogonal init) or
 * MRL_NET_CONV_ONLY (models.py:222-249: tf.contrib convolution2d, SAME padding, ReLU, xavier-uniform init).
 * Parameters: ONE flat f32 buffer per network copy (online / target) in TF variable-creation order; tensor i is
 * described by mrl_qnet_tensor_info (TF names "deepq/q_func/..."; init_kind 0 zeros, 1 orthogonal * init_scale,
 * 2 xavier uniform, 3 ones).  `update_target` (build_graph.py:423-428) is a device copy of that buffer by the caller. */
typedef struct mrl_qnet_desc {
    int network;            /* MRL_NET_MLP | MRL_NET_NATURE_CNN | MRL_NET_CONV_ONLY */
    int ob_ndim;  int ob_shape[3];  int ob_dtype;
    int num_layers, num_hidden, activation;      /* mlp */
    int nconv;  int convs[4][3];                 /* conv_only: (num_outputs, kernel_size, stride) per layer */
    int nhidden;  int hiddens[4];                /* build_q_func(hiddens=[256]) */
    int dueling;                                 /* build_q_func(dueling=True) */
    int nact;
    int layer_norm;                              /* build_q_func(layer_norm=True): LayerNorm before the ReLU of the hidden head layers */
    int body_layer_norm;                         /* network = mlp(layer_norm=True) (common/models.py:97-98): LayerNorm[_i]/beta, gamma
                                                    behind every mlp_fc<i> of the BODY, between the affine map and its activation */
} mrl_qnet_desc;
typedef struct mrl_qnet mrl_qnet;
int  mrl_qnet_create(const mrl_qnet_desc* desc, mrl_qnet** out);
void mrl_qnet_destroy(mrl_qnet* q);
long mrl_qnet_num_params(const mrl_qnet* q);
int  mrl_qnet_num_tensors(const mrl_qnet* q);
int  mrl_qnet_tensor_info(const mrl_qnet* q, int i, char* name, int name_cap, int* ndim, int shape[4], long* offset,
                          int* init_kind, double* init_scale);
size_t mrl_qnet_workspace_bytes(const mrl_qnet* q, int batch);
/* q_values (build_graph.py:441): q_out f32 [n][nact] */
int mrl_qnet_values(const mrl_qnet* q, const float* params, const void* obs, int n, float* q_out, void* workspace,
                    size_t workspace_bytes, int batch, void* stream);
/* act (build_graph.py:146-199): argmax_a q(obs) (first maximum), replaced by rand_actions[b] where uniforms[b] < eps
 * (uniforms == NULL: deterministic); n <= batch; q_out may be NULL */
int mrl_qnet_act(const mrl_qnet* q, const float* params, const void* obs, int n, float eps, const float* uniforms,
                 const int32_t* rand_actions, int32_t* actions_out, float* q_out, void* workspace,
                 size_t workspace_bytes, int batch, void* stream);
/* parameter-space noise (build_graph.py:202-315): mean over n rows of KL(softmax(q_a) || softmax(q_b)), q_* f32 [n][nact] --
 * the action-space distance between the unperturbed and an adaptively perturbed network that steers the noise scale
 * (:283-291).  Perturbed parameter copies are ordinary flat parameter buffers passed to mrl_qnet_values / mrl_qnet_act. */
int mrl_qnet_policy_kl(const float* q_a, const float* q_b, int n, int nact, float* mean_kl_out, void* stream);
/* train, gradient part (build_graph.py:380-413): td_out f32 [B] = q_t(s,a) - (r + gamma (1-done) q_tp1_best),
 * loss_out f32 [1] = mean(w * huber(td)), grads_out f32 [P] = d loss / d online params (un-clipped) */
int mrl_qnet_td_grad(const mrl_qnet* q, const float* params, const float* target_params, const void* obs_t,
                     const int32_t* act, const float* rew, const void* obs_tp1, const float* done, const float* weights,
                     float gamma, int double_q, int B, float* grads_out, float* td_out, float* loss_out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* train, optimizer part (build_graph.py:416-421, deepq.py:205): every variable's gradient clipped by ITS OWN norm
 * (tf.clip_by_norm; grad_norm_clipping <= 0: none), then TF-1 ApplyAdam with alpha = lr*sqrt(1-b2^t)/(1-b1^t).
 * alpha_dev != NULL: the step size is read from device memory instead of `alpha` (launch graphs that are replayed
 * with a step size that changes from step to step) */
int mrl_qnet_adam_step(const mrl_qnet* q, float* params, float* grads, float* adam_m, float* adam_v, float alpha,
                       const float* alpha_dev, float beta1, float beta2, float eps, float grad_norm_clipping,
                       void* workspace, size_t workspace_bytes, int batch, void* stream);

/* ---- synthetic device-resident VecEnv (bench/test data source) ----------------------------
 * Stands where gym environments stand in the reference (common/vec_env/): lock-step stepping with
 * AUTO-RESET on done (subproc_vec_env.py:8-12) and Monitor-style episode bookkeeping
 * (bench/monitor.py:58-77).  Dynamics = counter-based integer hash (spec in csrc/envs.hip), bit-
 * reproducible by the NumPy twin SyntheticVecEnvCPU.  State: ep u32 [N], st i32 [N], ep_ret f32 [N].
 * step: consumes actions (int32 [N] when discrete, else ignored), advances the state, writes the
 * NEXT observations (reset obs where done), rewards, dones and, where an episode finished, its
 * return/length (0 elsewhere; fin_* may both be NULL). */
int mrl_synth_env_obs(uint32_t seed, int ob_elems, int ob_u8, int N, const uint32_t* ep,
                      const int32_t* st, void* obs_out, void* stream);
int mrl_synth_env_step(uint32_t seed, int ob_elems, int ob_u8, int discrete, int reward_kind,
                       int lmin, int lspan, int N, uint32_t* ep, int32_t* st, float* ep_ret,
                       const int32_t* actions, void* obs_out, float* rew_out, uint8_t* done_out,
                       float* fin_r_out, int32_t* fin_l_out, void* stream);

/* ---- actor-side wrappers on the device -----------------------------------
 * K11 VecFrameStack.step_wait / reset --- common/vec_env/vec_frame_stack.py:17-30.  stacked [N][pix][S],
 * obs [N][pix][C], elements of `esize` bytes (1 or 4); in place.  Reference semantics: the stack axis is
 * rolled left by ONE element per step (np.roll shift=-1: a frame shift when C == 1), envs with news != 0
 * are zeroed (reset != 0: all of them), the last C elements become the new observation. */
int mrl_framestack_step(void* stacked, const void* obs, const uint8_t* news, int N, long pix, int S, int C,
                        int esize, int reset, void* stream);
/* K12 VecNormalize --- common/vec_env/vec_normalize.py:26-47, common/running_mean_std.py:12-33.
 * ob: batch mean/var per column in float32, row after row (== np.mean/np.var(axis=0), bit-exact), merged
 * into the float64 running mean[D] / var[D] (count = samples seen BEFORE this batch, host-tracked), then
 * out = clip((obs - mean) / sqrt(var + epsilon), +-clipob) computed in float64; out_f64 may be NULL.
 * rew: ret[N] (f64 state) = ret*gamma + rews; rms[2] = {mean, var} of ret updated the same way;
 * out = clip(rews / sqrt(var + epsilon), +-cliprew); ret[news] = 0. */
int mrl_vecnorm_ob(const float* obs, int N, int D, double* mean, double* var, double count, double epsilon,
                   double clipob, float* out_f32, double* out_f64, void* stream);
int mrl_vecnorm_rew(const float* rews, const uint8_t* news, int N, double* ret, double* rms, double count,
                    double gamma, double epsilon, double cliprew, float* out_f32, double* out_f64,
                    void* stream);

/* ---- optional HIP-event profiler ------------------------------------------------------------
 * When enabled every kernel launch of the library is bracketed by hipEvents recorded on the launch
 * stream, accumulated per label ("c1.fwd", "c2.wgrad", "heads", "clip+adam", ...) together with
 * the ALGORITHMIC flops / bytes of the launch.  Reading a label synchronises.
 * Off by default.  (Process-global state of the library: this table, the engine options and the tile-tuning overrides below.) */
int mrl_prof_enable(int on);           /* on != 0: clear counters and start; 0: stop */
int mrl_prof_num_labels(void);
int mrl_prof_get(int i, char* name, int name_cap, long* count, double* total_ms,
                 double* total_flops, double* total_bytes);

/* ---- tile-shape tuning hook (bench / profiling only) ------------------------------------------
 * Overrides the GEMM tile variant used for one launch site, label = "<layer>.<fwd|wgrad|dgrad>"
 * (e.g. "c1.fwd"); variant < 0 restores the built-in choice.  Results are identical for every
 * variant of the forward / data-gradient GEMMs; weight-gradient split-K changes summation order. */
int mrl_tune_set(const char* label, int variant);
/* Engine options (process-wide; defaults also settable through the environment variable in brackets).  Every option
 * selects between engines that compute the SAME products (tests/test_gpu_kernels.py::test_engine_options_agree), except
 * "f32_bf16x6", which selects the arithmetic of the fp32 x fp32 GEMM sites:
 *   "f32_bf16x6"  [MRL_F32_BF16X6, 2]  fp32 x fp32 GEMM sites on the bf16 pipe with both operands split EXACTLY into 3 bf16
 *                  planes (round-to-nearest at each level) and the six plane products x_i w_j, i + j <= 2, accumulated in
 *                  fp32: what is dropped is at most 2^-24 of a product -- the rounding of one IEEE fp32 multiply -- and
 *                  3.5e-9 of it on average (csrc/wres.hip.h, tests/test_split_arithmetic.py); 0 = fp32 MFMA (bitwise fmaf
 *                  chain).  Builds with -DMRL_PRODUCTS8 keep two more products (< 2^-33) at 8/6 of the matrix time;
 *                  the read-only option "f32_products" reports which build is loaded (6 | 8)
 *   "u8_bf16x3"   [MRL_U8_BF16X3, 1]  first conv layer (uint8 pixels) on the bf16 pipe with an exact 3-way bf16 split of the
 *                  other operand; 0 = fp32 MFMA
 *   "tr_epilogue" [MRL_TR_EPILOGUE, 1]  split engines (hidden conv / fc forward, data gradients) accumulate
 *                  transposed (D^T = B A^T): a lane owns one output row, 16-byte stores, in-lane ReLU mask words;
 *                  0 = row-major accumulators, ballot mask words
 *   "relu_bits"   [MRL_RELU_BITS, 1]  conv forward epilogues also write a 1-bit-per-element ReLU mask that the data gradients
 *                  read instead of the fp32 activations; 0 = fp32 activations
 *   "dgrad_x6"    [MRL_DGRAD_X6, 2]  conv data gradients on the position-major tiled split engine: 2 = the product configuration
 *                  (transposed epilogue, bit-mask act', whole tiles) as a kernel of its own whose control flow gives the compiler
 *                  exact vmcnt counts, so that a tile's output stores drain while the next tile multiplies (dgrad_x6p_kernel,
 *                  round 6); 1 = the generic kernel for every case (round 5; what partial tiles still take); bit-identical.
 *                  0 = LDS-resident fp32-MFMA engine ("dgrad_async" [MRL_DGRAD_ASYNC, 1]: its LDS-DMA staging form for conv2)
 *   "wgrad_tr"    [MRL_WGRAD_TR, 1]  weight gradients of conv2 / conv3 / fc1 on the split-arithmetic transpose-read kernels
 *                  (wgradtr.hip.h: operands staged in their natural layout, split once, fragments fetched with LDS transpose
 *                  reads; needs f32_bf16x6 != 0); 0 = "wgrad_x8" / fp32-MFMA engines
 *   "wgrad_x8"    [MRL_WGRAD_X8, 1]  (wgrad_tr = 0) weight gradients on the transposed-staging tiles of wgradx8.hip.h: 1 = layers
 *                  with >= 128 outputs (fc1), 2 = conv2 / conv3 too, 0 = image-resident / tiled fp32-MFMA engines
 *   "c1_wgrad2"   [MRL_C1_WGRAD2, 3]  first conv layer weight gradient with both operands transposed while staged
 *                  (c1wgrad.hip.h): 3 = half-image work units, two workgroups per CU, operand loads two units ahead (two
 *                  register sets); 2 = the same, one unit ahead; 1 = whole images; 0 = per-byte gathers.  3 and 2 are
 *                  bit-identical.
 *   "c1_lds"      [MRL_C1_LDS, 4]  first conv layer forward on the image-resident engines: 4 = one software pipeline per wave
 *                  (the previous image's stores / mask words, the next image's conversion and LDS writes and the loads of the
 *                  one after it are issued between the MFMAs of the current image), transposed accumulators with 16-byte
 *                  stores; 3 = the same pipeline with row-major accumulators; 2 = the same arithmetic in lock-step phases
 *                  (multiply, epilogue, stage, barrier); 1 = uint8 images converted per fragment; 0 = weights-resident gather
 *                  engine.  3 and 2 are bit-identical (same products, same order of accumulation); 4 (round 6) spreads the twelve whole 32-pixel tiles
 *                  of an image three to a SIMD and multiplies the pixels 384 .. 399 as two 16 x 16 blocks on v_mfma_f32_16x16x32_bf16
 *                  (no padded tile on the busiest SIMD: 2.80 -> 2.50 ms per 131072 images): the same exact products added in groups of
 *                  32 instead of 16 for those pixels, within 1e-6 of 3 / 2.
 *   "x6_pg"       [MRL_X6_PG, 8]  tile order of the tiled split engines: row panels of an XCD that advance through the column
 *                  tiles together (a weight tile pulled into that XCD's L2 serves x6_pg row panels); 1 = one panel at a time
 *   "gae_lane"    [MRL_GAE_LANE, 1]  GAE(lambda) with one environment per lane from 64 environments up (gae_lane_kernel); 0 = the
 *                  LDS-staged 16-environments-per-workgroup kernel at every size.  Bit-identical (and to the reference's Runner.run).
 *   "mlp_act"     [MRL_MLP_ACT, 1]  act side of the 2 x 64 tanh MLP nets on float observations: both layers of both nets in ONE launch
 *                  (csrc/mlpact.hip.h); 0 = layer by layer on the tiled GEMM.  Agree to 2e-6 (different order of the fp32 sums).
 *   "x6_splitk"   [MRL_X6_SPLITK, 1]  act-side fc launches of the tiled split engine whose tiles alone leave half the chip idle (fc1 of
 *                  NatureCNN at num_envs <= 8192): K split over blockIdx.y into partial slabs + the bias / activation pass; 0 = one
 *                  workgroup walks all of K.
 *   "wgrad_xcd"   [MRL_WGRAD_XCD, 1]  fc weight gradient (wgrad_tr): XCD x takes a contiguous run of the (sample slab, tile) list, so the
 *                  workgroups that read the same rows share an L2 (fc1: counter traffic 4.3 -> 2.1 GB per launch); 0 = launch order.
 *                  Bit-identical.
 *   "dqn_overlap" [MRL_DQN_OVERLAP, 1]  mrl_qnet_td_grad: the online network's obs_t / obs_tp1 passes as one batch when the caller hands
 *                  them over back to back, the target network's pass on a side stream (needs a workspace of
 *                  mrl_qnet_workspace_bytes(2 B) + mrl_qnet_workspace_bytes(B)); 0 = three passes one after the other.
 *   "dqn_heads"   [MRL_DQN_HEADS, 1]  the standard dueling heads (one hidden layer per head, no layer norm, hidden width % 32 == 0,
 *                  batch <= 256) as four fp32-MFMA kernels (csrc/qheads.hip.h) instead of 7 + 13 layer-wise launches; 0 = layer by layer.
 *   "dqn_latdgrad" [MRL_DQN_LATDGRAD, 1]  both heads' first layers into the latent in one launch (q_lat_dgrad_kernel); 0 = two tiled GEMMs.
 *   "dqn_wstream" [MRL_DQN_WSTREAM, 1]  mrl_qnet_td_grad: weight gradients (off the dz chain) on side streams next to the data
 *                  gradients, consecutive layers round-robin over three streams with their own split-K scratch (2 / 3: one / two
 *                  streams); 0 = everything on the caller's stream.
 *   "dqn_pair"    [MRL_DQN_PAIR, 1]  mrl_qnet_td_grad with obs_t | obs_tp1 back to back and a conv_only body: the online pass (2 B rows)
 *                  and the target pass (B rows) go through the SAME launches (every kernel takes two problems); 0 = two passes.
 *   "conv_skinny" [MRL_CONV_SKINNY, 1]  conv forward / data gradient on the generic path at latency-bound sizes (<= 2 x CUs 128-row tiles):
 *                  register-direct skinny-tile kernels (csrc/convskinny.hip.h), and row splits of 64 instead of 256 for the tiled weight
 *                  gradient; 0 = the tiled engine as before.
 *   "conv_splitk" [MRL_CONV_SPLITK, 1]  hidden conv layers on the generic tiled engine at small batches (the Q-network's conv2 / conv3 at
 *                  batch 32-64: a few dozen workgroups walking K alone): K split over the z dimension into partial slabs + the bias /
 *                  activation pass; 0 = one workgroup per output tile walks all of K.
 *   "x6_ktm"      [MRL_X6_KTM, 1]  weight planes of the fc layers' tiled split launches (fc1 forward / data gradient) in k-tile-major
 *                  order [plane][k / 32][n][k % 32]: a staging load of 16 rows x 64 bytes touches 8 whole cache lines instead of
 *                  16 half lines (round 6); 0 = [plane][n][k].  Bit-identical.
 *   "conv_x6c"    [MRL_CONV_X6C, 1]  conv2 / conv3 forward of NatureCNN from 96 images up on the class-resident kernel
 *                  (convx6c.hip.h): a tile is whole images, the input pixels of one stride-parity class are staged once as raw
 *                  fp32 and serve all taps of the class (every input element loaded once, 4 / 2 barrier pairs per tile instead of
 *                  16 / 18), operand split on the fragment path between the MFMAs, weight fragments as coalesced 1 KB loads;
 *                  0 = tiled engine.  Same products, k walked in class-major order.
 *   "x6_frag"     [MRL_X6_FRAG, 1]  tiled split engine, 64-filter conv forward launches (the batches "conv_x6c" does not take): the
 *                  streamed operand is staged into LDS as raw fp32 and split into its three bf16 planes on the FRAGMENT path,
 *                  between the MFMAs of the wave that owns the rows (gemmx6r.hip.h), k walked in class-major order;
 *                  0 = split while staging, natural k order (gemmx6.hip.h, the round-4 engine).
 *   "wgrad_pipe"  [MRL_WGRAD_PIPE, 1]  conv2 / conv3 weight gradients (wgrad_tr): the next image's operand split runs between the
 *                  MFMAs of the current image and its planes wait in registers for the barrier; 0 = a split phase of its own.
 *                  Bit-identical.
 *   "x6_dither"   [MRL_X6_DITHER, 3]  two bits.  Bit 0: the tiled split engines (forward and data-gradient GEMMs) stage every other
 *                  group of 8 rows of their streamed operand with its sign flipped and undo that in the epilogue (no extra instructions): the small
 *                  bias toward -inf that v_mfma_f32_32x32x16_bf16 has for products far below its accumulator then alternates
 *                  from sample to sample instead of adding up in the sums over a minibatch; the weight-gradient
 *                  engines do the same with every other partial slab.  Bit 1: the tiled engines' staging threads take the rows of
 *                  a group of 8 in the order 0 4 1 5 2 6 3 7, which keeps their LDS stores free of bank conflicts (bit-identical,
 *                  c2.fwd / c3.fwd -1.5 .. -2 %).  0 = everything staged as is
 *   "fused_norm"  [MRL_FUSED_NORM, 1]  mrl_model_train_step takes the global norm from the gradient reductions
 *   "lstm_e1"     [MRL_LSTM_E1, 1]  LSTM scans of long trajectories with ONE environment per workgroup (plain fmaf chains, two
 *                  barriers per step) when groups of four environments would leave CUs idle; 0 = always four per workgroup
 *                  (4x4x1 MFMA form).  Bit-identical.
 *   "mlp_fused"   [MRL_MLP_FUSED, 1]  whole-step kernel for the 2 x 64 tanh MLP; 0 = layer-wise launches
 *   "mlp_waves"   [MRL_MLP_WAVES, 8]  waves per workgroup of that kernel (8 | 4)
 *   "mlp_slice"   [MRL_MLP_SLICE, 1]  separate policy / value nets: one workgroup per (32-sample tile, net) instead of one per
 *                  tile -- half the serial chain per workgroup, 256 instead of 128 workgroups for a 4096-sample minibatch
 *   "heads_wave"  [MRL_HEADS_WAVE, 2]  loss / head-gradient kernel for the NatureCNN head shape: 2 = two samples per wave-step (lane half =
 *                  sample for the loss algebra; <= 6 actions, else as 1), 1 = one sample per wave-step (bit-identical), 0 = generic tile kernel.
 *                  Any non-zero value also selects the ACT side's wave-per-sample heads kernel (heads_act_wave_kernel: the same
 *                  dot-product form; actions identical, values / neglogp within 1e-6 of the tile kernel's)
 * Builds with -DMRL_X6_EXPERIMENTS (MRL_BUILD_DEFINES, csrc/build.py) add the measured-and-dropped variants that
 * profiles/README.md and scripts/ab_options.py refer to ("act_planes", "x6_il", "x6_spec", "*_dbg", ...); they are not part of
 * the product library.  Returns MRL_EINVAL for unknown names.  mrl_get_option reports the value in effect. */
int mrl_set_option(const char* name, int value);
int mrl_get_option(const char* name, int* value_out);

#ifdef __cplusplus
}
#endif
#endif /* MRL_H_ */
