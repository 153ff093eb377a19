// linear.hip -- the large Linear layers of the PPO update (forward, input gradient, weight/bias gradient) on gfx950 in
// the same split-precision bf16x3 MFMA arithmetic as the rollout forward (gemm3.h).
// Reference: the torch.nn.Linear calls of SpatialEdgeSelfAttn.forward / spatial_linear
// (rl/networks/selfAttn_srnn_temp_node.py:63-91,408) executed under autograd by PPO.update (rl/ppo.py:60-95:
// evaluate_actions -> loss.backward()).
#include "common.h"
#include "gemm3.h"

namespace {

// GRU cell, pointwise part, for the PPO update (torch.nn.GRU gate order r,z,n; rl/networks/srnn_model.py:35-105 runs the
// cell per time step with the hidden state multiplied by the done mask first).  gi = x W_ih^T + b_ih, gh = hm W_hh^T + b_hh
// (hm = masked previous hidden state).  The forward keeps (r, z, n, gh_n) for the backward.
__global__ __launch_bounds__(128) void gru_cell_fwd_kernel(int N, const float *__restrict__ gi, const float *__restrict__ gh, const float *__restrict__ hm,
                                                           float *__restrict__ h_out, float *__restrict__ gates)
{
    const int e = blockIdx.x, c = threadIdx.x;
    const float *gie = gi + (size_t)e * 384, *ghe = gh + (size_t)e * 384;
    const float hn = ghe[256 + c];
    const float r = 1.0f / (1.0f + expf(-(gie[c] + ghe[c])));
    const float z = 1.0f / (1.0f + expf(-(gie[128 + c] + ghe[128 + c])));
    const float n = tanhf(gie[256 + c] + r * hn);
    h_out[(size_t)e * 128 + c] = (1.0f - z) * n + z * hm[(size_t)e * 128 + c];
    float *g = gates + (size_t)e * 512;
    g[c] = r; g[128 + c] = z; g[256 + c] = n; g[384 + c] = hn;
}

// dh' -> d(gi) [N,384], d(gh) [N,384] and the direct path d(hm) = dh' * z
__global__ __launch_bounds__(128) void gru_cell_bwd_kernel(int N, const float *__restrict__ gates, const float *__restrict__ hm,
                                                           const float *__restrict__ dh, float *__restrict__ dgi, float *__restrict__ dgh,
                                                           float *__restrict__ dhm)
{
    const int e = blockIdx.x, c = threadIdx.x;
    const float *g = gates + (size_t)e * 512;
    const float r = g[c], z = g[128 + c], n = g[256 + c], hn = g[384 + c];
    const float d = dh[(size_t)e * 128 + c], h = hm[(size_t)e * 128 + c];
    const float din = d * (1.0f - z) * (1.0f - n * n); // d(i_n)
    const float dr = din * hn * r * (1.0f - r);          // d(i_r) = d(h_r)
    const float dz = d * (h - n) * z * (1.0f - z);       // d(i_z) = d(h_z)
    float *a = dgi + (size_t)e * 384, *b = dgh + (size_t)e * 384;
    a[c] = dr; a[128 + c] = dz; a[256 + c] = din;
    b[c] = dr; b[128 + c] = dz; b[256 + c] = din * r;
    dhm[(size_t)e * 128 + c] = d * z;
}

// embedding_layer.0 of the human-human block in the update: Linear(D -> 128) + ReLU with D = 2 or 12 input features
// (selfAttn_srnn_temp_node.py:33-36).  K is far too small for the matrix cores and its weight gradient is a reduction of
// several hundred thousand rows into a [128, D] matrix: thread n owns output column n; rows are walked grid-stride
// (coalesced 512-byte row reads of y / dy, broadcast reads of the D inputs); per-block partial sums are reduced in
// block order afterwards (deterministic).
__global__ __launch_bounds__(128) void embed0_fwd_kernel(int R, int D, const float *__restrict__ x, const float *__restrict__ W,
                                                         const float *__restrict__ b, float *__restrict__ y)
{
    const int n = threadIdx.x;
    float w[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) w[d] = d < D ? W[n * D + d] : 0.0f;
    const float bn = b[n];
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        const float *xr = x + (size_t)r * D;
        float acc = bn;
#pragma unroll
        for (int d = 0; d < 16; ++d)
            if (d < D) acc += xr[d] * w[d];
        y[(size_t)r * 128 + n] = fmaxf(acc, 0.0f);
    }
}

__global__ __launch_bounds__(128) void embed0_bwd_kernel(int R, int D, const float *__restrict__ x, const float *__restrict__ y,
                                                         const float *__restrict__ dy, float *__restrict__ part)
{
    const int n = threadIdx.x;
    float acc[17];
#pragma unroll
    for (int d = 0; d < 17; ++d) acc[d] = 0.0f;
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        const float g = y[(size_t)r * 128 + n] > 0.0f ? dy[(size_t)r * 128 + n] : 0.0f;
        const float *xr = x + (size_t)r * D;
#pragma unroll
        for (int d = 0; d < 16; ++d)
            if (d < D) acc[d] += g * xr[d];
        acc[16] += g;
    }
    float *p = part + ((size_t)blockIdx.x * 128 + n) * (D + 1); // [block][n][D weights | bias]
#pragma unroll
    for (int d = 0; d < 16; ++d)
        if (d < D) p[d] = acc[d];
    p[D] = acc[16];
}

} // namespace

extern "C" int cn_embed0_fwd(int R, int D, const float *x, const float *W, const float *b, float *y, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(R >= 1 && D >= 1 && D <= 16 && x && W && b && y, "cn_embed0_fwd: bad argument (D must be in [1,16])");
    const int blocks = R < 8192 ? R : 8192;
    hipLaunchKernelGGL(embed0_fwd_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, R, D, x, W, b, y);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_embed0_bwd(int R, int D, const float *x, const float *y, const float *dy, int blocks, float *partials, float *dWb, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(R >= 1 && D >= 1 && D <= 16 && x && y && dy && partials && dWb && blocks >= 1, "cn_embed0_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (blocks > R) blocks = R;
    hipLaunchKernelGGL(embed0_bwd_kernel, dim3(blocks), dim3(128), 0, st, R, D, x, y, dy, partials);
    CN_CHECK_LAUNCH();
    const size_t n = (size_t)128 * (D + 1);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, blocks, partials, dWb);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_gru_cell_fwd(int N, const float *gi, const float *gh, const float *hm, float *h_out, float *gates, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(N >= 1 && gi && gh && hm && h_out && gates, "cn_gru_cell_fwd: bad argument");
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, N, gi, gh, hm, h_out, gates);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_gru_cell_bwd(int N, const float *gates, const float *hm, const float *dh, float *dgi, float *dgh, float *dhm, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(N >= 1 && gates && hm && dh && dgi && dgh && dhm, "cn_gru_cell_bwd: bad argument");
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, N, gates, hm, dh, dgi, dgh, dhm);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_split_bf16(const float *w, int rows, int cols, int transpose, void *hi, void *lo, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(w && hi && lo && rows > 0 && cols > 0, "cn_split_bf16: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)rows * cols;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (transpose) hipLaunchKernelGGL(split_bf16_t_kernel, dim3(blocks), dim3(256), 0, st, rows, cols, w, (__bf16 *)hi, (__bf16 *)lo);
    else hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks), dim3(256), 0, st, n, w, (__bf16 *)hi, (__bf16 *)lo);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_linear_fwd(int M, int N, int K, const float *X, int ldx, const float *relu_gate, const void *Whi, const void *Wlo, const float *bias,
                             int act, float *Y, int ldy, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(X && Whi && Wlo && Y && M >= 0, "cn_linear_fwd: bad argument");
    CN_REQUIRE(act == 0 || act == 1, "cn_linear_fwd: act must be 0 (none) or 1 (relu)");
    CN_REQUIRE(ldx >= K && ldy >= N, "cn_linear_fwd: leading dimension smaller than the row length");
    hipStream_t st = (hipStream_t)stream;
    if (act == 1) return launch_gemm3<128, ACT_RELU>(M, N, K, X, ldx, (const __bf16 *)Whi, (const __bf16 *)Wlo, bias, Y, ldy, st, nullptr, relu_gate);
    return launch_gemm3<128, ACT_NONE>(M, N, K, X, ldx, (const __bf16 *)Whi, (const __bf16 *)Wlo, bias, Y, ldy, st, nullptr, relu_gate);
}

extern "C" int cn_linear_wgrad_splits(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0 || N % 64 || K % 128) return 0;
    const long long tiles = (long long)((N + 127) / 128) * (K / 128);
    long long s = (768 + tiles - 1) / tiles;             // 1.5 resident rounds of blocks on 256 CUs (shorter blocks, small tail) ...
    if (s > 64) s = 64;                                  // ... but bounded: every split costs an [N,K] partial to write and re-read
    const long long chunks = ((long long)M + BK3 - 1) / BK3;
    if (s > chunks) s = chunks;
    return (int)(s < 1 ? 1 : s);
}

extern "C" int cn_linear_wgrad(int M, int N, int K, const float *dY, int ldy, const float *relu_gate, const float *X, int ldx, int splits,
                               float *partials, float *db_partials, float *dW, float *db, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(dY && X && partials && dW && M > 0 && splits >= 1, "cn_linear_wgrad: bad argument");
    CN_REQUIRE(N % 64 == 0 && K % 128 == 0, "cn_linear_wgrad: N=%d must be a multiple of 64 and K=%d of 128", N, K);
    CN_REQUIRE((db == nullptr) == (db_partials == nullptr), "cn_linear_wgrad: db and db_partials go together");
    CN_REQUIRE(ldy >= N && ldx >= K, "cn_linear_wgrad: leading dimension smaller than the row length");
    hipStream_t st = (hipStream_t)stream;
    int rows = (M + splits - 1) / splits;
    rows = (rows + BK3 - 1) / BK3 * BK3;
    const int used = (M + rows - 1) / rows; // <= splits, every split non-empty
    constexpr size_t lds = (size_t)(2 * BM + 2 * 128) * L3_STRIDE * sizeof(__bf16);
    static bool attr_set = false;
    if (!attr_set) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm3_tn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm3_tn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    if (relu_gate) hipLaunchKernelGGL(gemm3_tn_kernel<true>, dim3((N + 127) / 128, K / 128, used), dim3(256), lds, st, M, N, K, dY, ldy, relu_gate, X, ldx, rows, partials, db_partials);
    else hipLaunchKernelGGL(gemm3_tn_kernel<false>, dim3((N + 127) / 128, K / 128, used), dim3(256), lds, st, M, N, K, dY, ldy, relu_gate, X, ldx, rows, partials, db_partials);
    CN_CHECK_LAUNCH();
    const size_t nk = (size_t)N * K;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, nk, used, partials, dW);
    CN_CHECK_LAUNCH();
    if (db) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, (size_t)N, used, db_partials, db);
        CN_CHECK_LAUNCH();
    }
    return CN_OK;
}
