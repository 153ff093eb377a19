// ppo.hip -- the pointwise / reduction part of PPO.update on gfx950: clipped surrogate + clipped value loss forward and
// backward, global gradient-norm clip and the Adam step over ONE flat parameter bucket.
// Reference: rl/ppo/ppo.py:66-84 (losses), :86-93 (zero_grad, backward, nn.utils.clip_grad_norm_, optimizer.step with
// torch.optim.Adam(lr, eps), ppo.py:32).
//
// All reductions are two-level and order-fixed (block partials in fp64 summed in block order), so results do not depend on
// scheduling.  Tie rules follow torch's autograd: min/max of two equal operands send half of the gradient to each,
// clamp passes the gradient on the closed interval.
#include "common.h"

#include <cmath>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PPO_BLOCKS = 256; // partial-sum slots of the loss reduction
constexpr int NORM_BLOCKS = 1024; // partial-sum slots of the gradient-norm reduction

struct LossTerm { float vl, al, dv, dlp; };

// One sample of ppo.py:66-84.  dv / dlp are d(value_loss_i)/d(values_i) and d(-min(surr1, surr2)_i)/d(logp_i), unscaled.
__device__ __forceinline__ LossTerm ppo_term(float v, float lp, float old_lp, float adv, float vp, float ret, float clip, bool clipped_value)
{
    LossTerm t;
    const float ratio = expf(lp - old_lp);
    const float s1 = ratio * adv;
    const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
    const float s2 = rc * adv;
    t.al = -fminf(s1, s2);
    const float g1 = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);   // torch.min backward: ties split evenly
    const float in_r = (ratio >= 1.0f - clip && ratio <= 1.0f + clip) ? 1.0f : 0.0f;
    t.dlp = -(g1 * s1 + (1.0f - g1) * in_r * s1);                 // d(ratio)/d(lp) = ratio
    const float e = v - ret;
    if (clipped_value) {
        const float d = v - vp;
        const float dc = fminf(fmaxf(d, -clip), clip);
        const float ec = (vp + dc) - ret;
        const float a = e * e, b = ec * ec;
        t.vl = 0.5f * fmaxf(a, b);
        const float ga = a > b ? 1.0f : (a == b ? 0.5f : 0.0f);
        const float in_v = (d >= -clip && d <= clip) ? 1.0f : 0.0f;
        t.dv = ga * e + (1.0f - ga) * in_v * ec;                  // 0.5 * 2 * (...)
    } else {
        t.vl = 0.5f * e * e;
        t.dv = e;
    }
    return t;
}

__global__ __launch_bounds__(256) void ppo_loss_partial_kernel(int64_t n, const float *__restrict__ values, const float *__restrict__ logp,
                                                               const float *__restrict__ old_logp, const float *__restrict__ adv,
                                                               const float *__restrict__ value_preds, const float *__restrict__ returns, float clip,
                                                               int clipped_value, double *__restrict__ partials)
{
    __shared__ double red[2][4];
    double sv = 0.0, sa = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const LossTerm t = ppo_term(values[i], logp[i], old_logp[i], adv[i], value_preds[i], returns[i], clip, clipped_value != 0);
        sv += (double)t.vl; sa += (double)t.al;
    }
    sv = wv_sum(sv); sa = wv_sum(sa);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sv; red[1][threadIdx.x >> 6] = sa; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ void ppo_loss_final_kernel(int64_t n, int blocks, const double *__restrict__ partials, float *__restrict__ losses)
{
    // one wavefront: lane l adds partials l, l + 64, ... (ascending), then a fixed butterfly -- deterministic, and four dependent loads deep
    // instead of 256 (the serial loop of one lane took 22 us per optimiser step)
    double sv = 0.0, sa = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 64) { sv += partials[2 * b]; sa += partials[2 * b + 1]; }
    sv = wv_sum(sv); sa = wv_sum(sa);
    if (threadIdx.x == 0) {
        losses[0] = (float)(sv / (double)n); // value_loss = 0.5 * mean(max(...))
        losses[1] = (float)(sa / (double)n); // action_loss = -mean(min(surr1, surr2))
    }
}

__global__ __launch_bounds__(256) void ppo_loss_bwd_kernel(int64_t n, const float *__restrict__ values, const float *__restrict__ logp,
                                                           const float *__restrict__ old_logp, const float *__restrict__ adv,
                                                           const float *__restrict__ value_preds, const float *__restrict__ returns, float clip,
                                                           int clipped_value, const float *__restrict__ g_losses, float *__restrict__ d_values,
                                                           float *__restrict__ d_logp)
{
    const float inv_n = 1.0f / (float)n;
    const float gv = g_losses[0] * inv_n, ga = g_losses[1] * inv_n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const LossTerm t = ppo_term(values[i], logp[i], old_logp[i], adv[i], value_preds[i], returns[i], clip, clipped_value != 0);
        d_values[i] = gv * t.dv;
        d_logp[i] = ga * t.dlp;
    }
}

// sum of squares of the flat gradient bucket (scaled by grad_scale, e.g. 1 / world_size after a sum all-reduce)
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(int64_t n, const float *__restrict__ grad, float grad_scale, double *__restrict__ partials)
{
    __shared__ double red[4];
    double s = 0.0;
    const int64_t n4 = n >> 2;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(grad);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 g = g4[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float x = g[q] * grad_scale; s += (double)(x * x); }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) { const float x = grad[(n4 << 2) + threadIdx.x] * grad_scale; s += (double)(x * x); }
    s = wv_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// nn.utils.clip_grad_norm_ (clip_coef = max_norm / (total_norm + 1e-6), clamped to 1) followed by torch.optim.Adam's
// single-tensor update (no weight decay, no amsgrad), in torch's operation order:
//   exp_avg.lerp_(g, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2);
//   denom = exp_avg_sq.sqrt() / sqrt(1 - b2^t) + eps; p.addcdiv_(exp_avg, denom, value = -lr / (1 - b1^t))
// Every block re-derives the total norm from the partials in block order (a few KB from L2; keeps it one launch).
__global__ __launch_bounds__(256) void adam_step_kernel(int64_t n, float *__restrict__ param, float *__restrict__ grad, float *__restrict__ exp_avg,
                                                        float *__restrict__ exp_avg_sq, const double *__restrict__ partials, int n_partials,
                                                        float grad_scale, float max_norm, float lr_over_bc1, float w1, float beta2, float w2,
                                                        float bc2_sqrt, float eps, float *__restrict__ norm_out)
{
    __shared__ float coef_s;
    if (threadIdx.x < 64) {
        double s = 0.0;
        for (int b = threadIdx.x; b < n_partials; b += 64) s += partials[b];
        // fixed lane order tree: deterministic for a given n_partials
        s = wv_sum(s);
        if (threadIdx.x == 0) {
            const float total = (float)sqrt(s);
            float c = grad_scale;
            if (max_norm > 0.0f) {
                const float cc = max_norm / (total + 1e-6f);
                c *= cc < 1.0f ? cc : 1.0f;
            }
            coef_s = c;
            if (blockIdx.x == 0 && norm_out) norm_out[0] = total;
        }
    }
    __syncthreads();
    const float coef = coef_s;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float g = grad[i] * coef;
        grad[i] = g; // clip_grad_norm_ scales .grad in place
        const float m0 = exp_avg[i];
        const float m = m0 + w1 * (g - m0);           // lerp_ with weight < 0.5
        const float v = exp_avg_sq[i] * beta2 + w2 * g * g;
        exp_avg[i] = m; exp_avg_sq[i] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        param[i] = param[i] - lr_over_bc1 * (m / denom);
    }
}

} // namespace

extern "C" int cn_ppo_loss_workspace_doubles(void) { return 2 * PPO_BLOCKS; }

extern "C" int cn_ppo_loss_fwd(int64_t n, const float *values, const float *logp, const float *old_logp, const float *adv,
                               const float *value_preds, const float *returns, float clip_param, int use_clipped_value_loss,
                               double *workspace, float *losses, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(n > 0 && values && logp && old_logp && adv && value_preds && returns && workspace && losses, "cn_ppo_loss_fwd: bad argument");
    int blocks = (int)((n + 255) / 256);
    if (blocks > PPO_BLOCKS) blocks = PPO_BLOCKS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ppo_loss_partial_kernel, dim3(blocks), dim3(256), 0, st, n, values, logp, old_logp, adv, value_preds, returns, clip_param,
                       use_clipped_value_loss, workspace);
    CN_CHECK_LAUNCH();
    hipLaunchKernelGGL(ppo_loss_final_kernel, dim3(1), dim3(64), 0, st, n, blocks, workspace, losses);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_ppo_loss_bwd(int64_t n, const float *values, const float *logp, const float *old_logp, const float *adv,
                               const float *value_preds, const float *returns, float clip_param, int use_clipped_value_loss,
                               const float *g_losses, float *d_values, float *d_logp, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(n > 0 && values && logp && old_logp && adv && value_preds && returns && g_losses && d_values && d_logp, "cn_ppo_loss_bwd: bad argument");
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ppo_loss_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, values, logp, old_logp, adv, value_preds, returns,
                       clip_param, use_clipped_value_loss, g_losses, d_values, d_logp);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_adam_workspace_doubles(void) { return NORM_BLOCKS; }

extern "C" int cn_adam_clip_step(int64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, double grad_scale, double max_grad_norm,
                                 double lr, double beta1, double beta2, double eps, int64_t step, double *workspace, float *grad_norm_out,
                                 void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(n > 0 && param && grad && exp_avg && exp_avg_sq && workspace, "cn_adam_clip_step: bad argument");
    CN_REQUIRE(step >= 1, "cn_adam_clip_step: step counts from 1");
    CN_REQUIRE(((uintptr_t)grad & 15) == 0, "cn_adam_clip_step: grad must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)(((n >> 2) + 255) / 256);
    nb = nb < 1 ? 1 : (nb > NORM_BLOCKS ? NORM_BLOCKS : nb);
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(nb), dim3(256), 0, st, n, grad, (float)grad_scale, workspace);
    CN_CHECK_LAUNCH();
    // bias corrections like torch (python floats -> double), handed to the kernel as float scalars
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(256), 0, st, n, param, grad, exp_avg, exp_avg_sq, workspace, nb, (float)grad_scale, (float)max_grad_norm,
                       (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)std::sqrt(bc2), (float)eps, grad_norm_out);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
