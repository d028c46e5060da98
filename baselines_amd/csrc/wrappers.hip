// Actor-side VecEnv wrappers on the device: observations produced by a device-resident
// environment are frame-stacked / normalised without leaving HBM.
//
// Reference map (paths relative to baselines/):
//   VecFrameStack.step_wait / reset      common/vec_env/vec_frame_stack.py:17-30
//   VecNormalize._obfilt / step_wait     common/vec_env/vec_normalize.py:26-47
//   RunningMeanStd.update                common/running_mean_std.py:12-33
// HBM-bound elementwise / column-reduction kernels (no MFMA).  The batch statistics reproduce NumPy's
// float32 axis-0 reduction order (row after row) so the observation path is bit-exact; the 1-D float64
// reduction of the returns uses a fixed tree (NumPy: pairwise), equal to ~1e-16 relative.
#include <math.h>

#include "common.hip.h"

using namespace mrl;

namespace {

// every pixel keeps S consecutive elements (the stacked channel axis).  Reference semantics: roll the axis left
// by ONE element (np.roll shift=-1; a frame shift only when C == 1), zero finished envs, overwrite the last C.
template <typename E>
__global__ __launch_bounds__(256) void framestack_kernel(E* __restrict__ stacked, const E* __restrict__ obs,
                                                         const uint8_t* __restrict__ news, long N, long pix, int S,
                                                         int C, int reset) {
    const long total = N * pix;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += (long)gridDim.x * 256L) {
        const long n = q / pix;
        E* s = stacked + q * S;
        const E* o = obs + q * C;
        const bool zero = reset || (news && news[n]);
        for (int k = 0; k < S - C; ++k) s[k] = zero ? (E)0 : s[k + 1];
        for (int k = 0; k < C; ++k) s[S - C + k] = o[k];
    }
}

// per-column batch statistics in float32, row-after-row like np.mean / np.var(axis=0), then the float64
// parallel-variance merge of running_mean_std.py:22-33
__global__ __launch_bounds__(64) void vecnorm_stats_kernel(const float* __restrict__ x, int N, int D,
                                                           double* __restrict__ mean, double* __restrict__ var,
                                                           double count) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= D) return;
    float s = 0.f;
    for (int r = 0; r < N; ++r) s = __fadd_rn(s, x[(long)r * D + c]);
    const float m32 = __fdiv_rn(s, (float)N);
    float s2 = 0.f;
    for (int r = 0; r < N; ++r) {
        const float d = __fsub_rn(x[(long)r * D + c], m32);
        s2 = __fadd_rn(s2, __fmul_rn(d, d));
    }
    const float v32 = __fdiv_rn(s2, (float)N);
    const double bc = (double)N;
    const double delta = (double)m32 - mean[c];
    const double tot = count + bc;
    const double new_mean = mean[c] + delta * bc / tot;
    // batch_var (float32 array) * batch_count (Python int) stays float32 in NumPy before it meets the float64 state
    const double m_a = var[c] * count, m_b = (double)__fmul_rn(v32, (float)N);
    const double M2 = m_a + m_b + delta * delta * count * bc / tot;
    mean[c] = new_mean;
    var[c] = M2 / tot;
}

__global__ __launch_bounds__(256) void vecnorm_apply_kernel(const float* __restrict__ x, long total, int D,
                                                            const double* __restrict__ mean,
                                                            const double* __restrict__ var, double eps, double clip,
                                                            float* __restrict__ out32, double* __restrict__ out64) {
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += (long)gridDim.x * 256L) {
        const int c = (int)(q % D);
        double v = ((double)x[q] - mean[c]) / sqrt(var[c] + eps);
        v = fmin(fmax(v, -clip), clip);
        if (out32) out32[q] = (float)v;
        if (out64) out64[q] = v;
    }
}

// returns: ret = ret*gamma + rew; running stats of ret (fixed-order f64 tree); scaled + clipped rewards; ret[news] = 0
__global__ __launch_bounds__(256) void vecnorm_rew_kernel(const float* __restrict__ rew, const uint8_t* __restrict__ news,
                                                          int N, double* __restrict__ ret, double* __restrict__ rms,
                                                          double count, double gamma, double eps, double clip,
                                                          float* __restrict__ out32, double* __restrict__ out64) {
    __shared__ double sh[4];
    __shared__ double s_mean, s_var;
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double r = ret[i] * gamma + (double)rew[i];
        ret[i] = r;
        s += r;
    }
    const double tsum = block_sum_256(s, sh);
    if (tid == 0) s_mean = tsum / (double)N;
    __syncthreads();
    const double bm = s_mean;
    double s2 = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double d = ret[i] - bm;
        s2 += d * d;
    }
    const double t2 = block_sum_256(s2, sh);
    if (tid == 0) {
        const double bv = t2 / (double)N, bc = (double)N;
        const double delta = bm - rms[0];
        const double tot = count + bc;
        const double new_mean = rms[0] + delta * bc / tot;
        const double M2 = rms[1] * count + bv * bc + delta * delta * count * bc / tot;
        rms[0] = new_mean;
        rms[1] = M2 / tot;
        s_var = M2 / tot;
    }
    __syncthreads();
    const double denom = sqrt(s_var + eps);
    for (int i = tid; i < N; i += 256) {
        double v = (double)rew[i] / denom;
        v = fmin(fmax(v, -clip), clip);
        if (out32) out32[i] = (float)v;
        if (out64) out64[i] = v;
        if (news[i]) ret[i] = 0.0;
    }
}

}  // namespace

extern "C" int mrl_framestack_step(void* stacked, const void* obs, const uint8_t* news, int N, long pix, int S, int C,
                                   int esize, int reset, void* stream) {
    if (!stacked || !obs || N <= 0 || pix <= 0 || S <= 0 || C <= 0 || C > S) return MRL_EINVAL;
    if (!reset && !news) return MRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * pix;
    const int blocks = (int)std::min<long>((total + 255) / 256, 16384);
    ProfScope ps("framestack", 0.0, (double)total * esize * (2.0 * S + C), st);
    if (esize == 1)
        hipLaunchKernelGGL(framestack_kernel<uint8_t>, dim3(blocks), dim3(256), 0, st, (uint8_t*)stacked, (const uint8_t*)obs, news, (long)N, pix, S, C, reset);
    else if (esize == 4)
        hipLaunchKernelGGL(framestack_kernel<uint32_t>, dim3(blocks), dim3(256), 0, st, (uint32_t*)stacked, (const uint32_t*)obs, news, (long)N, pix, S, C, reset);
    else
        return MRL_EUNSUP;
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrl_vecnorm_ob(const float* obs, int N, int D, double* mean, double* var, double count, double epsilon,
                              double clipob, float* out_f32, double* out_f64, void* stream) {
    if (!obs || !mean || !var || N <= 0 || D <= 0 || (!out_f32 && !out_f64)) return MRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("vecnorm_ob", 0.0, (double)N * D * (3.0 * 4 + 4), st);
    hipLaunchKernelGGL(vecnorm_stats_kernel, dim3((D + 63) / 64), dim3(64), 0, st, obs, N, D, mean, var, count);
    MRL_LAUNCH_CHECK();
    const long total = (long)N * D;
    const int blocks = (int)std::min<long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(vecnorm_apply_kernel, dim3(blocks), dim3(256), 0, st, obs, total, D, mean, var, epsilon, clipob, out_f32, out_f64);
    MRL_LAUNCH_CHECK();
    return 0;
}

extern "C" int mrl_vecnorm_rew(const float* rews, const uint8_t* news, int N, double* ret, double* rms, double count,
                               double gamma, double epsilon, double cliprew, float* out_f32, double* out_f64,
                               void* stream) {
    if (!rews || !news || !ret || !rms || N <= 0 || (!out_f32 && !out_f64)) return MRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("vecnorm_rew", 0.0, (double)N * 30.0, st);
    hipLaunchKernelGGL(vecnorm_rew_kernel, dim3(1), dim3(256), 0, st, rews, news, N, ret, rms, count, gamma, epsilon,
                       cliprew, out_f32, out_f64);
    MRL_LAUNCH_CHECK();
    return 0;
}
