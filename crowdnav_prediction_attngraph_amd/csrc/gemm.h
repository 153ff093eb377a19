// gemm.h -- exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32) shared by policy.hip and gst.hip.
#pragma once
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2,
       // backward epilogues (training path, rn_train in policy.hip): the product is a gradient w.r.t. an activation whose forward VALUE y is
       // handed over in GemmBatch::resid -- multiply by relu'(y) = [y > 0] or tanh'(.) = 1 - y^2 instead of adding a residual
       ACT_MUL_DRELU = 3, ACT_MUL_DTANH = 4 };

// XCD-aware tile mapping.  Workgroups are dispatched round-robin over the 8 XCDs (linear id L -> XCD L % 8), each with a
// private L2.  With the natural (x = column tile fastest) order the N/BN workgroups that share one A row-tile land on
// 8 different XCDs and every L2 fetches that tile again (rocprof: FETCH_SIZE 5-9x the algorithmic bytes).  Remap so
// that XCD x owns row tiles {x, x+8, ...} and walks their column tiles consecutively: q = L / 8 -> (row = (q / nbx) * 8
// + x, col = q % nbx).  gridDim.y must be a multiple of 8 (the launcher pads; surplus row tiles exit on the M check).
__device__ __forceinline__ void xcd_tile(int &row_tile, int &col_tile)
{
    const int nbx = gridDim.x;
    const int L = blockIdx.y * nbx + blockIdx.x;
    const int x = L & 7, q = L >> 3;
    col_tile = q % nbx;
    row_tile = (q / nbx) * 8 + x;
}
// Variant for wide weight matrices: when the whole W (hi + lo planes) plus the A tiles in flight exceed one XCD's 4 MB L2
// (q|k|v: 3.1 MB of W + 5 x 256 KB of A), the L2 thrashes and every A tile is fetched ~4x.  Here XCDs 0-3 take the left
// half of the column tiles and XCDs 4-7 the right half (W working set per L2 halves), XCD x' of a half owns row tiles
// x', x'+4, ...; an A tile is then fetched by two L2s instead of being re-fetched by one.  gridDim.x must be even.
__device__ __forceinline__ void xcd_tile_split(int &row_tile, int &col_tile)
{
    const int nbx = gridDim.x, ch = nbx >> 1;
    const int L = blockIdx.y * nbx + blockIdx.x;
    const int x = L & 7, q = L >> 3;
    col_tile = (x >> 2) * ch + q % ch;
    row_tile = (q / ch) * 4 + (x & 3);
}

constexpr int BM = 128, BK = 32, LDS_STRIDE = 36; // 36 floats = 144 B rows: conflict-free ds_read_b128

// C[M,N] (ldc) = act(A[M,K] (lda) * W[N,K]^T + bias[N]); N % BN == 0, K % 32 == 0, M arbitrary.
// 256 threads = 4 wavefronts in a 2x2 arrangement; each wavefront owns a 64 x (BN/2) tile = 2 x (BN/64) MFMA blocks.
// K order inside a group of 8 is remapped so each lane feeds 4 consecutive MFMA steps from ONE 16-byte LDS read:
// lanes 0-31 hold k = 8g+s, lanes 32-63 hold k = 8g+4+s at step s (same mapping for A and W, so the sum is unchanged).
// m_dev (optional): device-side row count (<= M).  The HH block runs on the compacted set of detected humans whose
// size is only known on the device; the grid is sized for the worst case and surplus row tiles exit immediately.
// blockIdx.z batches independent problems that share the shape (strides in floats; e.g. actor.2 / critic.2).
// relu_from: columns >= relu_from get a ReLU on top of ACT (lets one launch produce [t_emb | relu(enc)]).
struct GemmBatch { long long sA, sW, sB, sC; const float *resid; int ldr; }; // resid: optional residual added after the activation

template <int TBM, int BN, int ACT>
__global__ __launch_bounds__(256) void gemm_nt_kernel(int M, int N, int K, const float *__restrict__ A, int lda,
                                                      const float *__restrict__ W, const float *__restrict__ bias,
                                                      float *__restrict__ C, int ldc, const int *__restrict__ m_dev,
                                                      GemmBatch gb, int relu_from)
{
    if (m_dev) { const int md = *m_dev; M = md < M ? md : M; }
    int row_tile, col_tile;
    xcd_tile(row_tile, col_tile);
    if (row_tile * TBM >= M) return;
    A += blockIdx.z * gb.sA; W += blockIdx.z * gb.sW; C += blockIdx.z * gb.sC;
    if (bias) bias += blockIdx.z * gb.sB;
    constexpr int MI = TBM / 64;       // 32-row MFMA blocks per wavefront
    constexpr int NB = BN / 64;        // MFMA column blocks per wavefront
    constexpr int ALD = TBM / 32;      // float4 loads of A per thread per K tile
    constexpr int WLD = BN / 32;       // float4 loads of W per thread per K tile
    __shared__ __attribute__((aligned(16))) float As[TBM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Ws[BN * LDS_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_blk = row_tile * TBM, n_blk = col_tile * BN;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4; // staging: 8 lanes cover one 128-byte row segment

    f32x16 acc[MI][NB];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 pa[ALD], pw[WLD];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int p = 0; p < ALD; ++p) {
            const int r = m_blk + lrow + 32 * p;
            if (r < M) pa[p] = *reinterpret_cast<const f32x4 *>(A + (size_t)r * lda + k0 + lcol);
            else pa[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < WLD; ++p) {
            const int r = n_blk + lrow + 32 * p;
            pw[p] = *reinterpret_cast<const f32x4 *>(W + (size_t)r * K + k0 + lcol);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int p = 0; p < ALD; ++p) *reinterpret_cast<f32x4 *>(&As[(lrow + 32 * p) * LDS_STRIDE + lcol]) = pa[p];
#pragma unroll
        for (int p = 0; p < WLD; ++p) *reinterpret_cast<f32x4 *>(&Ws[(lrow + 32 * p) * LDS_STRIDE + lcol]) = pw[p];
    };

    load_tiles(0);
    const int half = lane >> 5, l31 = lane & 31;
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads(); // previous tile fully consumed
        store_tiles();
        __syncthreads();
        if (k0 + BK < K) load_tiles(k0 + BK); // prefetch next tile into registers while this one is multiplied
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 af[MI], bf[NB];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = *reinterpret_cast<const f32x4 *>(&As[(wm * (TBM / 2) + i * 32 + l31) * LDS_STRIDE + g * 8 + half * 4]);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                bf[j] = *reinterpret_cast<const f32x4 *>(&Ws[(wn * (BN / 2) + j * 32 + l31) * LDS_STRIDE + g * 8 + half * 4]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = n_blk + wn * (BN / 2) + j * 32 + l31;
            const float b = bias ? bias[col] : 0.0f;
            const bool extra_relu = col >= relu_from;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m_blk + wm * (TBM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M) {
                    float v = acc[i][j][r] + b;
                    if (ACT == ACT_RELU || extra_relu) v = fmaxf(v, 0.0f);
                    if (ACT == ACT_TANH) v = tanhf(v);
                    if (ACT == ACT_MUL_DRELU) v = gb.resid[(size_t)row * gb.ldr + col] > 0.0f ? v : 0.0f;
                    else if (ACT == ACT_MUL_DTANH) { const float y = gb.resid[(size_t)row * gb.ldr + col]; v *= 1.0f - y * y; }
                    else if (gb.resid) v += gb.resid[(size_t)row * gb.ldr + col];
                    C[(size_t)row * ldc + col] = v;
                }
            }
        }
}


template <int TBM, int BN, int ACT>
static int launch_gemm_t(int M, int N, int K, const float *A, int lda, const float *W, const float *bias, float *C, int ldc, hipStream_t st,
                         const int *m_dev, int nbatch, GemmBatch gb, int relu_from)
{
    CN_REQUIRE(N % BN == 0 && K % BK == 0 && lda % 4 == 0, "gemm: unsupported shape M=%d N=%d K=%d lda=%d", M, N, K, lda);
    if (M == 0) return CN_OK;
    dim3 grid(N / BN, (((M + TBM - 1) / TBM) + 7) & ~7, nbatch); // rows padded to a multiple of 8 for the XCD mapping
    hipLaunchKernelGGL((gemm_nt_kernel<TBM, BN, ACT>), grid, dim3(256), 0, st, M, N, K, A, lda, W, bias, C, ldc, m_dev, gb, relu_from);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

} // namespace
