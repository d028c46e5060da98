// Device-resident synthetic vectorised environment (bench / test data source).
//
// It stands where gym environments stand in the reference (common/vec_env/*): it honours the
// VecEnv contract -- lock-step stepping, AUTO-RESET on done (subproc_vec_env.py:8-12,
// dummy_vec_env.py:51-54), per-episode return/length bookkeeping (bench/monitor.py:58-77) -- but
// its "dynamics" are a counter-based integer hash, so a NumPy twin
// (baselines_amd/common/vec_env/synthetic_vec_env.py: SyntheticVecEnvCPU) reproduces every byte
// and the oracle can be fed identical rollouts.  Observations are written straight into HBM
// (Atari-shaped u8 84x84x4 or MuJoCo-shaped f32 vectors) and never touch the host.
//
//   mix32      = murmur3 fmix32
//   key(e)     = mix32(seed * 0x9E3779B1 + e)
//   k2(e,ep,st)= mix32(key ^ mix32(ep * 0x632BE5AB + st))
//   obs word w = mix32(k2 + w * 0x9E3779B9):  u8 obs take the 4 bytes (little endian), f32 obs =
//                (word >> 8) * 2^-23 - 1   (exact in f32, range [-1, 1))
//   episode length L(e,ep) = lmin + mix32(key ^ (ep * 0x85EBCA77 + 0x1234567)) % lspan
//   reward hash hr = mix32((k2 ^ 0xA5A5A5A5) + a * 0x27D4EB2F)   (a = discrete action, 0 for Box)
//     kind 0 (Atari ClipReward range): -1 if hr < 0.05*2^32, +1 if hr >= 0.95*2^32, else 0
//     kind 1: (hr >> 8) * 2^-23 - 1;   kind 2: +1 (CartPole-like)
#include "common.hip.h"

using namespace mrl;

struct SynthCfg {
    uint32_t seed;
    int ob_elems;     // elements per observation
    int ob_u8;        // 1: uint8 obs, 0: f32
    int discrete;     // 1: int32 actions [N]; 0: continuous (ignored by the reward)
    int reward_kind;
    int lmin, lspan;
};

__host__ __device__ static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t env_key(uint32_t seed, int e) { return mix32(seed * 0x9E3779B1u + (uint32_t)e); }
__device__ __forceinline__ uint32_t env_k2(uint32_t key, uint32_t ep, uint32_t st) {
    return mix32(key ^ mix32(ep * 0x632BE5ABu + st));
}

// scalars: reward / done / episode bookkeeping for step t (state = (ep, st) BEFORE the step)
__global__ __launch_bounds__(256) void synth_scalar_kernel(SynthCfg c, int N, uint32_t* __restrict__ ep,
                                                           int32_t* __restrict__ st, float* __restrict__ ep_ret,
                                                           const int32_t* __restrict__ actions,
                                                           float* __restrict__ rew, uint8_t* __restrict__ done,
                                                           float* __restrict__ fin_r, int32_t* __restrict__ fin_l) {
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    uint32_t key = env_key(c.seed, e);
    uint32_t epi = ep[e];
    int32_t s = st[e];
    uint32_t k2 = env_k2(key, epi, (uint32_t)s);
    uint32_t a = (c.discrete && actions) ? (uint32_t)actions[e] : 0u;
    uint32_t hr = mix32((k2 ^ 0xA5A5A5A5u) + a * 0x27D4EB2Fu);
    float r;
    if (c.reward_kind == 0) r = hr < 214748365u ? -1.f : (hr >= 4080218931u ? 1.f : 0.f);
    else if (c.reward_kind == 1) r = (float)(hr >> 8) * (1.f / 8388608.f) - 1.f;
    else r = 1.f;
    int L = c.lmin + (int)(mix32(key ^ (epi * 0x85EBCA77u + 0x1234567u)) % (uint32_t)c.lspan);
    float ret = ep_ret[e] + r;
    int len = s + 1;
    bool d = len >= L;
    rew[e] = r;
    done[e] = d ? 1 : 0;
    if (fin_r) { fin_r[e] = d ? ret : 0.f; fin_l[e] = d ? len : 0; }
    if (d) { ep[e] = epi + 1; st[e] = 0; ep_ret[e] = 0.f; }
    else { st[e] = len; ep_ret[e] = ret; }
}

// uint8 frames whose word count per env is a multiple of 4 (Atari: 84 * 84 * 4 / 4 = 7056): one thread = 4 hash words = one 16-byte store,
// one division per 16 bytes instead of one per 4 (round 6: 59 -> ~30 us per env step of 4096 frames, 115 MB)
__global__ __launch_bounds__(256) void synth_obs_u8x16_kernel(SynthCfg c, int N, const uint32_t* __restrict__ ep,
                                                              const int32_t* __restrict__ st, uint4* __restrict__ obs) {
    const int groups = c.ob_elems / 16;                      // 16-byte groups per env
    const long total = (long)N * groups;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += (long)gridDim.x * 256L) {
        const int e = (int)(q / groups);
        const uint32_t w = (uint32_t)(q - (long)e * groups) * 4u;
        const uint32_t k2 = env_k2(env_key(c.seed, e), ep[e], (uint32_t)st[e]);
        uint4 v;
        v.x = mix32(k2 + w * 0x9E3779B9u);
        v.y = mix32(k2 + (w + 1u) * 0x9E3779B9u);
        v.z = mix32(k2 + (w + 2u) * 0x9E3779B9u);
        v.w = mix32(k2 + (w + 3u) * 0x9E3779B9u);
        obs[q] = v;
    }
}
static void launch_synth_obs(const SynthCfg& c, int N, const uint32_t* ep, const int32_t* st, void* obs, hipStream_t s);

// observations of the CURRENT state (ep, st): one thread = one hash word (4 B of u8 / one float)
__global__ __launch_bounds__(256) void synth_obs_kernel(SynthCfg c, int N, const uint32_t* __restrict__ ep,
                                                        const int32_t* __restrict__ st, void* __restrict__ obs) {
    const int words = c.ob_u8 ? (c.ob_elems + 3) / 4 : c.ob_elems;
    const long total = (long)N * words;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += (long)gridDim.x * 256L) {
        int e = (int)(q / words);
        uint32_t w = (uint32_t)(q - (long)e * words);
        uint32_t k2 = env_k2(env_key(c.seed, e), ep[e], (uint32_t)st[e]);
        uint32_t h = mix32(k2 + w * 0x9E3779B9u);
        if (c.ob_u8) {
            uint8_t* o = static_cast<uint8_t*>(obs) + (long)e * c.ob_elems + (long)w * 4;
            if ((int)(w * 4 + 3) < c.ob_elems && (c.ob_elems % 4 == 0)) {
                *reinterpret_cast<uint32_t*>(o) = h;
            } else {
                for (int b = 0; b < 4 && (int)(w * 4 + b) < c.ob_elems; ++b) o[b] = (uint8_t)(h >> (8 * b));
            }
        } else {
            static_cast<float*>(obs)[(long)e * c.ob_elems + w] = (float)(h >> 8) * (1.f / 8388608.f) - 1.f;
        }
    }
}

static void launch_synth_obs(const SynthCfg& c, int N, const uint32_t* ep, const int32_t* st, void* obs, hipStream_t s) {
    if (c.ob_u8 && c.ob_elems % 16 == 0 && (uintptr_t)obs % 16 == 0) {
        const long total = (long)N * (c.ob_elems / 16);
        const int blocks = (int)min((total + 255) / 256, (long)16384);
        hipLaunchKernelGGL(synth_obs_u8x16_kernel, dim3(blocks), dim3(256), 0, s, c, N, ep, st, static_cast<uint4*>(obs));
        return;
    }
    const long total = (long)N * (c.ob_u8 ? (c.ob_elems + 3) / 4 : c.ob_elems);
    const int blocks = (int)min((total + 255) / 256, (long)16384);
    hipLaunchKernelGGL(synth_obs_kernel, dim3(blocks), dim3(256), 0, s, c, N, ep, st, obs);
}

extern "C" int mrl_synth_env_obs(uint32_t seed, int ob_elems, int ob_u8, int N, const uint32_t* ep,
                                 const int32_t* st, void* obs_out, void* stream) {
    if (!ep || !st || !obs_out || N <= 0 || ob_elems <= 0) return MRL_EINVAL;
    SynthCfg c{seed, ob_elems, ob_u8, 0, 0, 1, 1};
    launch_synth_obs(c, N, ep, st, obs_out, (hipStream_t)stream);
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrl_synth_env_step(uint32_t seed, int ob_elems, int ob_u8, int discrete, int reward_kind, int lmin,
                                  int lspan, int N, uint32_t* ep, int32_t* st, float* ep_ret,
                                  const int32_t* actions, void* obs_out, float* rew_out, uint8_t* done_out,
                                  float* fin_r_out, int32_t* fin_l_out, void* stream) {
    if (!ep || !st || !ep_ret || !obs_out || !rew_out || !done_out || N <= 0 || ob_elems <= 0 || lmin < 1 ||
        lspan < 1)
        return MRL_EINVAL;
    if ((fin_r_out == nullptr) != (fin_l_out == nullptr)) return MRL_EINVAL;
    SynthCfg c{seed, ob_elems, ob_u8, discrete, reward_kind, lmin, lspan};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(synth_scalar_kernel, dim3((N + 255) / 256), dim3(256), 0, s, c, N, ep, st, ep_ret, actions,
                       rew_out, done_out, fin_r_out, fin_l_out);
    MRL_LAUNCH_CHECK();
    launch_synth_obs(c, N, ep, st, obs_out, s);
    MRL_LAUNCH_CHECK();
    return 0;
}
