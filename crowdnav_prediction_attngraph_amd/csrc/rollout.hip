// rollout.hip -- GAE scan and advantage normalisation (rl/networks/storage.py:123-132, rl/ppo/ppo.py:37-39).
//
// GAE has a serial dependence over T (30) and none over envs: one lane per env, lanes of a wavefront read
// consecutive envs of a [T][N] row -> fully coalesced; values stay in registers across the scan.  fp32 with torch's
// operation order (-ffp-contract=off), so it is bit-identical to the reference loop.
// Advantage statistics are wavefront-shuffle reductions accumulated in fp64 in a fixed order (bit-reproducible); the three
// partial sums are exposed so that data-parallel ranks can all-reduce them before normalising (global mean / unbiased std).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void gae_kernel(int T, int N, const float *__restrict__ rewards, const float *__restrict__ values,
                                                  const float *__restrict__ masks, float gamma, float gl, float *__restrict__ returns)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float gae = 0.0f;
    float v_next = values[(size_t)T * N + n];
    for (int t = T - 1; t >= 0; --t) {
        const float m = masks[(size_t)(t + 1) * N + n];
        const float v = values[(size_t)t * N + n];
        const float delta = rewards[(size_t)t * N + n] + gamma * v_next * m - v;
        gae = delta + gl * m * gae;
        returns[(size_t)t * N + n] = gae + v;
        v_next = v;
    }
}

// ONE block, fixed summation order: the statistics (and with them every normalised advantage) are bit-reproducible from run to
// run, which a resumed training run relies on (floating-point atomics across blocks would sum in arrival order).
__global__ __launch_bounds__(1024) void adv_stats_kernel(int64_t n, const float *__restrict__ returns, const float *__restrict__ values,
                                                         double *stats)
{
    __shared__ double part[2][16];
    double s = 0.0, ss = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const double a = (double)(returns[i] - values[i]); // fp32 subtraction like torch, accumulated in fp64
        s += a; ss += a * a;
    }
    s = wv_sum(s); ss = wv_sum(ss);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][w] = s; part[1][w] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 16; ++k) { a += part[0][k]; b += part[1][k]; }
        stats[0] = a; stats[1] = b; stats[2] = (double)n;
    }
}

__global__ __launch_bounds__(256) void adv_norm_kernel(int64_t n, const float *__restrict__ returns, const float *__restrict__ values,
                                                       const double *__restrict__ stats, float *__restrict__ adv)
{
    const double cnt = stats[2];
    const double mean = stats[0] / cnt;
    double var = (stats[1] - cnt * mean * mean) / (cnt - 1.0); // unbiased (torch.std default)
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, denom = (float)sqrt(var) + 1e-5f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        adv[i] = ((returns[i] - values[i]) - meanf) / denom;
}

// bench.Monitor's aggregate over one vec-env step (rl/networks/envs.py:70-73 wraps every env; train.py:180-182 collects info['episode']['r']):
// acc[0] += finished episodes, [1] += their returns, [2] += their lengths, [3..5] += timeouts / collisions / goals.  One workgroup, fixed
// summation order: the sums do not depend on scheduling (a resumed run reports the same statistics as the uninterrupted one).
__global__ __launch_bounds__(1024) void episode_stats_kernel(int E, const uint8_t *__restrict__ done, const uint8_t *__restrict__ info,
                                                             const double *__restrict__ ep_ret, const int32_t *__restrict__ ep_len, double *acc)
{
    __shared__ double part[6][16];
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int e = threadIdx.x; e < E; e += 1024) {
        if (!done[e]) continue;
        const int c = info[e];
        v[0] += 1.0; v[1] += ep_ret[e]; v[2] += (double)ep_len[e];
        v[3] += c == CN_INFO_TIMEOUT ? 1.0 : 0.0; v[4] += c == CN_INFO_COLLISION ? 1.0 : 0.0; v[5] += c == CN_INFO_REACHGOAL ? 1.0 : 0.0;
    }
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const double t = wv_sum(v[k]); if ((threadIdx.x & 63) == 0) part[k][w] = t; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double a = 0.0;
        for (int k = 0; k < 16; ++k) a += part[threadIdx.x][k];
        acc[threadIdx.x] += a;
    }
}

} // namespace

extern "C" int cn_episode_stats_update(int E, const uint8_t *done, const uint8_t *info, const double *ep_return, const int32_t *ep_len, double *acc,
                                       void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(E > 0 && done && info && ep_return && ep_len && acc, "cn_episode_stats_update: bad argument");
    hipLaunchKernelGGL(episode_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, E, done, info, ep_return, ep_len, acc);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_gae(int T, int N, const float *rewards, const float *values, const float *masks, double gamma, double lam,
                      float *returns, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(T > 0 && N > 0 && rewards && values && masks && returns, "cn_gae: bad argument");
    hipLaunchKernelGGL(gae_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, N, rewards, values, masks,
                       (float)gamma, (float)(gamma * lam), returns);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_adv_stats(int64_t n, const float *returns, const float *values, double *stats, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(n > 0 && returns && values && stats, "cn_adv_stats: bad argument");
    hipLaunchKernelGGL(adv_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, returns, values, stats);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_adv_normalize(int64_t n, const float *returns, const float *values, const double *stats, float *adv, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(n > 0 && returns && values && stats && adv, "cn_adv_normalize: bad argument");
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adv_norm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, returns, values, stats, adv);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
