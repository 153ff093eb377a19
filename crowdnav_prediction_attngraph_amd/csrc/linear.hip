// linear.hip -- the large Linear layers of the PPO update (forward, input gradient, weight/bias gradient) on gfx950 in
// the same split-precision bf16x3 MFMA arithmetic as the rollout forward (gemm3.h).
// Reference: the torch.nn.Linear calls of SpatialEdgeSelfAttn.forward / spatial_linear
// (rl/networks/selfAttn_srnn_temp_node.py:63-91,408) executed under autograd by PPO.update (rl/ppo.py:60-95:
// evaluate_actions -> loss.backward()).
#include "common.h"
#include "gemm3.h"

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

extern "C" int cn_linear_fwd(int M, int N, int K, const float *X, int ldx, const void *Whi, const void *Wlo, const float *bias, int act, float *Y,
                             int ldy, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(X && Whi && Wlo && Y && M >= 0, "cn_linear_fwd: bad argument");
    CN_REQUIRE(act == 0 || act == 1, "cn_linear_fwd: act must be 0 (none) or 1 (relu)");
    CN_REQUIRE(ldx >= K && ldy >= N, "cn_linear_fwd: leading dimension smaller than the row length");
    hipStream_t st = (hipStream_t)stream;
    if (act == 1) return launch_gemm3<128, ACT_RELU>(M, N, K, X, ldx, (const __bf16 *)Whi, (const __bf16 *)Wlo, bias, Y, ldy, st, nullptr);
    return launch_gemm3<128, ACT_NONE>(M, N, K, X, ldx, (const __bf16 *)Whi, (const __bf16 *)Wlo, bias, Y, ldy, st, nullptr);
}

extern "C" int cn_linear_wgrad_splits(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0 || N % 128 || K % 128) return 0;
    const long long tiles = (long long)(N / 128) * (K / 128);
    long long s = (768 + tiles - 1) / tiles;             // ~3 resident waves of blocks on 256 CUs
    const long long chunks = ((long long)M + BK3 - 1) / BK3;
    if (s > chunks) s = chunks;
    return (int)(s < 1 ? 1 : s);
}

extern "C" int cn_linear_wgrad(int M, int N, int K, const float *dY, int ldy, const float *X, int ldx, int splits, float *partials,
                               float *db_partials, float *dW, float *db, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(dY && X && partials && dW && M > 0 && splits >= 1, "cn_linear_wgrad: bad argument");
    CN_REQUIRE(N % 128 == 0 && K % 128 == 0, "cn_linear_wgrad: N=%d and K=%d must be multiples of 128", N, K);
    CN_REQUIRE((db == nullptr) == (db_partials == nullptr), "cn_linear_wgrad: db and db_partials go together");
    CN_REQUIRE(ldy >= N && ldx >= K, "cn_linear_wgrad: leading dimension smaller than the row length");
    hipStream_t st = (hipStream_t)stream;
    int rows = (M + splits - 1) / splits;
    rows = (rows + BK3 - 1) / BK3 * BK3;
    const int used = (M + rows - 1) / rows; // <= splits, every split non-empty
    constexpr size_t lds = (size_t)(2 * BM + 2 * 128) * L3_STRIDE * sizeof(__bf16);
    static bool attr_set = false;
    if (!attr_set) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm3_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm3_tn_kernel, dim3(N / 128, K / 128, used), dim3(256), lds, st, M, N, K, dY, ldy, X, ldx, rows, partials, db_partials);
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
