// gemm3.h -- split-precision ("bf16x3") MFMA GEMMs shared by policy.hip (rollout forward) and linear.hip (PPO update).
#pragma once
#include "common.h"
#include "gemm.h"

namespace {

// ---- split-precision GEMM: fp32 operands as (hi + lo) bf16 pairs, three bf16 MFMAs per product term ------------------
// a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with hi = bf16(x), lo = bf16(x - hi): the dropped terms are <= 2^-16 relative,
// accumulation is fp32 (measured end-to-end error on the HH block: 1.5e-5, bar 1e-4).  Runs on v_mfma_f32_32x32x16_bf16
// (16x the fp32 MFMA rate, three passes -> 5.3x).  A is fp32 in HBM and split while it is staged into LDS
// (v_cvt_pk_bf16_f32); W is split once per weight snapshot.  LDS rows are 32 bf16 padded to 40 (80 B): the 16-byte
// fragment reads of 16 consecutive rows then hit 16 distinct 16-B slots of the 256-B bank row (conflict-free).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// K tile = ONE MFMA k-step (16): 24.6 KB of LDS and <= 128 VGPRs per workgroup -> FOUR workgroups (16 wavefronts) per CU.
// The kernel is a plain two-barrier loop whose phases (global loads in flight, convert + LDS stores, MFMA, C stores) do
// not overlap inside one workgroup; what hides them is other workgroups in other phases, so occupancy beats tile depth:
// measured on the q|k|v shapes (stand-alone, random operands)  BK 64 / 2 per CU: 180 us | 2348 us (M = 24.5 k | 368 k),
// BK 32 / 3 per CU: 168 | 2025,  BK 16 / 4 per CU: 161 | 1888 (307 TFLOP/s algorithmic = 920 executed).
#ifndef CN_BK3
#define CN_BK3 16
#endif
constexpr int BK3 = CN_BK3;
constexpr int L3_STRIDE = BK3 + 8;  // +8 bf16 pad: rows 48 B apart (BK3 = 16) -> 16 consecutive rows' 16-byte reads tile all 64 banks
constexpr int G3_OCC = BK3 <= 16 ? 4 : (BK3 <= 32 ? 3 : 2); // workgroups per CU the register budget is compiled for

template <int TBM, int BN, int ACT, bool GATE>
__global__ __launch_bounds__(256, (GATE && G3_OCC > 2) ? G3_OCC - 1 : G3_OCC) void gemm3_nt_kernel(int M, int N, int K, const float *__restrict__ A, int lda,
                                                       const float *__restrict__ Agate, const __bf16 *__restrict__ Whi, const __bf16 *__restrict__ Wlo,
                                                       const float *__restrict__ bias, float *__restrict__ C, int ldc,
                                                       const int *__restrict__ m_dev)
{
    if (m_dev) { const int md = *m_dev; M = md < M ? md : M; }
    int row_tile, col_tile;
    if ((size_t)N * K > (size_t)512 * 1024 && !(gridDim.x & 1)) xcd_tile_split(row_tile, col_tile); // W planes > 2 MB: halve the per-L2 W set
    else xcd_tile(row_tile, col_tile);
    if (row_tile * TBM >= M) return;
    constexpr int MI = TBM / 64;             // 32-row MFMA blocks per wavefront (2 x 2 wavefronts: TBM/2 rows each)
    constexpr int NB = BN / 64;
    constexpr int AQ = BK3 / 4, ARP = 256 / AQ;  // float4 per A row of the K tile, rows staged per pass
    constexpr int ALD = TBM / ARP;               // float4 loads of A per thread per K tile
    constexpr int WQ = BK3 / 8;                  // 16-byte chunks per W row of the K tile
    constexpr int WCH = BN * WQ / 256;           // chunks of each W plane per thread per K tile
    static_assert(WCH >= 1 && ALD >= 1, "tile too small for 256 staging threads");
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    __bf16 *Ah = reinterpret_cast<__bf16 *>(smem3);
    __bf16 *Al = Ah + TBM * L3_STRIDE;
    __bf16 *Wh = Al + TBM * L3_STRIDE;
    __bf16 *Wl = Wh + BN * L3_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_blk = row_tile * TBM, n_blk = col_tile * BN;
    const int lrow = tid / AQ, lcol = (tid % AQ) * 4; // A staging: AQ lanes cover one row segment of the K tile

    f32x16 acc[MI][NB];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 pa[ALD], pg[GATE ? ALD : 1];
    bf16x8 pwh[WCH], pwl[WCH];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int p = 0; p < ALD; ++p) {
            const int r = m_blk + lrow + ARP * p;
            if (r < M) {
                pa[p] = *reinterpret_cast<const f32x4 *>(A + (size_t)r * lda + k0 + lcol);
                // backward through a ReLU: A = dY gated by the forward output (same shape / leading dimension).  Only the raw
                // load is issued here; the select happens in store_tiles so that it does not wait for the prefetch
                if (GATE) pg[p] = *reinterpret_cast<const f32x4 *>(Agate + (size_t)r * lda + k0 + lcol);
            } else {
                pa[p] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (GATE) pg[p] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int p = 0; p < WCH; ++p) {
            const int c = tid + 256 * p, r = n_blk + c / WQ, col = (c % WQ) * 8;
            pwh[p] = *reinterpret_cast<const bf16x8 *>(Whi + (size_t)r * K + k0 + col);
            pwl[p] = *reinterpret_cast<const bf16x8 *>(Wlo + (size_t)r * K + k0 + col);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int p = 0; p < ALD; ++p) {
            bf16x4 hi, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = GATE ? (pg[GATE ? p : 0][q] > 0.0f ? pa[p][q] : 0.0f) : pa[p][q];
                hi[q] = (__bf16)a;
                lo[q] = (__bf16)(a - (float)hi[q]);
            }
            *reinterpret_cast<bf16x4 *>(&Ah[(lrow + ARP * p) * L3_STRIDE + lcol]) = hi;
            *reinterpret_cast<bf16x4 *>(&Al[(lrow + ARP * p) * L3_STRIDE + lcol]) = lo;
        }
#pragma unroll
        for (int p = 0; p < WCH; ++p) {
            const int c = tid + 256 * p, r = c / WQ, col = (c % WQ) * 8;
            *reinterpret_cast<bf16x8 *>(&Wh[r * L3_STRIDE + col]) = pwh[p];
            *reinterpret_cast<bf16x8 *>(&Wl[r * L3_STRIDE + col]) = pwl[p];
        }
    };

    load_tiles(0);
    const int half = lane >> 5, l31 = lane & 31;
    for (int k0 = 0; k0 < K; k0 += BK3) {
        __syncthreads();
        store_tiles();
        __syncthreads();
        if (k0 + BK3 < K) load_tiles(k0 + BK3);
#pragma unroll
        for (int ks = 0; ks < BK3 / 16; ++ks) {
            bf16x8 ah[MI], al[MI], bh[NB], bl[NB];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int o = (wm * (TBM / 2) + i * 32 + l31) * L3_STRIDE + ks * 16 + half * 8;
                ah[i] = *reinterpret_cast<const bf16x8 *>(&Ah[o]);
                al[i] = *reinterpret_cast<const bf16x8 *>(&Al[o]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int o = (wn * (BN / 2) + j * 32 + l31) * L3_STRIDE + ks * 16 + half * 8;
                bh[j] = *reinterpret_cast<const bf16x8 *>(&Wh[o]);
                bl[j] = *reinterpret_cast<const bf16x8 *>(&Wl[o]);
            }
            // term-major issue order: consecutive MFMAs hit different accumulators (no back-to-back dependent chain)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    // Epilogue.  The bias is fetched and waited for once, and a tile that lies inside M stores without per-row predicates: with
    // the predicate every store sits in its own basic block behind an s_waitcnt vmcnt(0) (the compiler cannot tell there that
    // the bias load has landed), i.e. every store waits for the previous one to complete (see gemm3p.h).
    float bv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[j] = bias ? bias[n_blk + wn * (BN / 2) + j * 32 + l31] : 0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    auto store_all = [&](auto guard) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int col = n_blk + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m_blk + wm * (TBM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = acc[i][j][r] + bv[j];
                    if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
                    if (ACT == ACT_TANH) v = tanhf(v);
                    // streaming result (150 MB for q|k|v, far beyond any L2): non-temporal stores keep A / W resident in the L2
                    if (guard(row)) __builtin_nontemporal_store(v, &C[(size_t)row * ldc + col]);
                }
            }
    };
    if (m_blk + TBM <= M) store_all([](int) { return true; });
    else store_all([&](int row) { return row < M; });
}

// split a fp32 weight matrix into bf16 hi / lo parts
__global__ void split_bf16_kernel(size_t n, const float *__restrict__ w, __bf16 *__restrict__ hi, __bf16 *__restrict__ lo)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const __bf16 h = (__bf16)w[i];
        hi[i] = h;
        lo[i] = (__bf16)(w[i] - (float)h);
    }
}

template <int TBM, int BN, int ACT>
static int launch_gemm3_t(int M, int N, int K, const float *A, int lda, const __bf16 *Whi, const __bf16 *Wlo, const float *bias, float *C, int ldc,
                          hipStream_t st, const int *m_dev, const float *Agate = nullptr)
{
    CN_REQUIRE(N % BN == 0 && K % BK3 == 0 && lda % 4 == 0, "gemm3: unsupported shape M=%d N=%d K=%d lda=%d", M, N, K, lda);
    if (M == 0) return CN_OK;
    dim3 grid(N / BN, (((M + TBM - 1) / TBM) + 7) & ~7);
    constexpr size_t lds = (size_t)(2 * TBM + 2 * BN) * L3_STRIDE * sizeof(__bf16); // 73.7 KB at 128 x 128: needs the opt-in above 64 KB
    static CnLdsOptIn opt_in; // per device
    int opt_dev;
    if (opt_in.needed(&opt_dev)) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm3_nt_kernel<TBM, BN, ACT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm3_nt_kernel<TBM, BN, ACT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        opt_in.done(opt_dev);
    }
    if (Agate) hipLaunchKernelGGL((gemm3_nt_kernel<TBM, BN, ACT, true>), grid, dim3(256), lds, st, M, N, K, A, lda, Agate, Whi, Wlo, bias, C, ldc, m_dev);
    else hipLaunchKernelGGL((gemm3_nt_kernel<TBM, BN, ACT, false>), grid, dim3(256), lds, st, M, N, K, A, lda, Agate, Whi, Wlo, bias, C, ldc, m_dev);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
template <int BN, int ACT>
static int launch_gemm3(int M, int N, int K, const float *A, int lda, const __bf16 *Whi, const __bf16 *Wlo, const float *bias, float *C, int ldc,
                        hipStream_t st, const int *m_dev, const float *Agate = nullptr)
{
    return launch_gemm3_t<BM, BN, ACT>(M, N, K, A, lda, Whi, Wlo, bias, C, ldc, st, m_dev, Agate);
}

// hi/lo split of W^T: w [rows, cols] row-major -> hi, lo [cols, rows].  Lets the NT kernel compute dX = dY * W
// (the "weight" operand of that product is W^T).  Sizes are a few hundred KB: no tiling needed.
__global__ void split_bf16_t_kernel(int rows, int cols, const float *__restrict__ w, __bf16 *__restrict__ hi, __bf16 *__restrict__ lo)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // index into the transposed output
    if (i < (size_t)rows * cols) {
        const int c = (int)(i / rows), r = (int)(i % rows);
        const float x = w[(size_t)r * cols + c];
        const __bf16 h = (__bf16)x;
        hi[i] = h;
        lo[i] = (__bf16)(x - (float)h);
    }
}

// Weight-gradient GEMM (TN): P[s][n][k] = sum_{m in split s} dY[m][n] * X[m][k], both operands fp32 activations with the
// reduction index m as the SLOW axis in memory.  The MFMA wants 8 consecutive reduction elements per lane, so the tiles
// are transposed on their way into LDS: thread (column c, group g) loads 8 rows m of its column with 8 coalesced dword
// loads (64 lanes = 256 contiguous bytes each), splits them into bf16 hi/lo and writes ONE 16-byte LDS word per plane
// at [c][8g .. 8g+7].  Consecutive lanes hit rows 144 B apart -> conflict-free ds_write_b128, and the LDS image is
// exactly the NT kernel's, so the MFMA section is shared.  The m range is cut into `gridDim.z` splits (partials summed
// by reduce_partials_kernel in a fixed order: deterministic).  Blocks of k tile 0 also produce the column sums of dY
// (the bias gradient) from the registers they stage anyway.
template <bool GATE>
__global__ __launch_bounds__(256, (GATE && G3_OCC > 2) ? G3_OCC - 1 : G3_OCC) void gemm3_tn_kernel(int M, int N, int K, const float *__restrict__ dY, int ldy, const float *__restrict__ Ygate,
                                                       const float *__restrict__ X, int ldx, int rows_per_split, float *__restrict__ partials,
                                                       float *__restrict__ db_part)
{
    constexpr int BN = 128;
    constexpr int NB = BN / 64;
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    __bf16 *Ah = reinterpret_cast<__bf16 *>(smem3);
    __bf16 *Al = Ah + BM * L3_STRIDE;
    __bf16 *Wh = Al + BM * L3_STRIDE;
    __bf16 *Wl = Wh + BN * L3_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_blk = blockIdx.x * BM, k_blk = blockIdx.y * BN, split = blockIdx.z;
    const int m_begin = split * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    const int c = tid & 127, g0 = tid >> 7;
    const bool n_ok = n_blk + c < N; // N may end inside the tile (64-wide layers): the surplus columns stay zero
    const bool want_db = db_part != nullptr && blockIdx.y == 0;

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    constexpr int TG = BK3 / 16; // 8-row groups per thread per chunk (2 thread halves x TG groups x 8 rows = BK3 rows)
    float pa[TG][8], pb[TG][8], pg[GATE ? TG : 1][8];
    const float *a_col = dY + n_blk + c, *b_col = X + k_blk + c;
    const float *g_col = GATE ? Ygate + n_blk + c : nullptr; // backward through a ReLU: dY gated by the forward output
    auto load_chunk = [&](int m0) {
#pragma unroll
        for (int p = 0; p < TG; ++p)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = m0 + (g0 + 2 * p) * 8 + u;
                const bool ok = m < m_end;
                pa[p][u] = ok && n_ok ? a_col[(size_t)m * ldy] : 0.0f;
                if (GATE) pg[p][u] = ok && n_ok ? g_col[(size_t)m * ldy] : 0.0f; // raw load; the select happens in store_chunk
                pb[p][u] = ok ? b_col[(size_t)m * ldx] : 0.0f;
            }
    };
    float colsum = 0.0f;
    auto store_chunk = [&]() {
#pragma unroll
        for (int p = 0; p < TG; ++p) {
            bf16x8 ahi, alo, bhi, blo;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (GATE) pa[p][u] = pg[GATE ? p : 0][u] > 0.0f ? pa[p][u] : 0.0f;
                ahi[u] = (__bf16)pa[p][u];
                alo[u] = (__bf16)(pa[p][u] - (float)ahi[u]);
                bhi[u] = (__bf16)pb[p][u];
                blo[u] = (__bf16)(pb[p][u] - (float)bhi[u]);
                colsum += pa[p][u];
            }
            const int o = c * L3_STRIDE + (g0 + 2 * p) * 8;
            *reinterpret_cast<bf16x8 *>(&Ah[o]) = ahi;
            *reinterpret_cast<bf16x8 *>(&Al[o]) = alo;
            *reinterpret_cast<bf16x8 *>(&Wh[o]) = bhi;
            *reinterpret_cast<bf16x8 *>(&Wl[o]) = blo;
        }
    };

    load_chunk(m_begin);
    const int half = lane >> 5, l31 = lane & 31;
    for (int m0 = m_begin; m0 < m_end; m0 += BK3) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (m0 + BK3 < m_end) load_chunk(m0 + BK3);
#pragma unroll
        for (int ks = 0; ks < BK3 / 16; ++ks) {
            bf16x8 ah[2], al[2], bh[NB], bl[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = (wm * 64 + i * 32 + l31) * L3_STRIDE + ks * 16 + half * 8;
                ah[i] = *reinterpret_cast<const bf16x8 *>(&Ah[o]);
                al[i] = *reinterpret_cast<const bf16x8 *>(&Al[o]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int o = (wn * (BN / 2) + j * 32 + l31) * L3_STRIDE + ks * 16 + half * 8;
                bh[j] = *reinterpret_cast<const bf16x8 *>(&Wh[o]);
                bl[j] = *reinterpret_cast<const bf16x8 *>(&Wl[o]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    float *P = partials + (size_t)split * N * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = k_blk + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n_blk + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < N) P[(size_t)row * K + col] = acc[i][j][r];
            }
        }
    if (want_db) { // uniform per block
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem3);
        red[tid] = colsum;
        __syncthreads();
        if (tid < 128 && n_blk + tid < N) db_part[(size_t)split * N + n_blk + tid] = red[tid] + red[tid + 128];
    }
}

// out[i] = sum_s part[s][i] in split order (deterministic)
__global__ void reduce_partials_kernel(size_t n, int splits, const float *__restrict__ part, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float acc = 0.0f;
        for (int s = 0; s < splits; ++s) acc += part[(size_t)s * n + i];
        out[i] = acc;
    }
}

// The same sum for MANY partials of a SMALL output (embed0's 128 x (D + 1) gradient from thousands of blocks, a bias gradient from
// 64 splits): one output per wavefront instead of per thread -- lane l adds partials l, l + 64, ... (ascending), then the 64 lane
// sums are combined in a fixed butterfly order: deterministic, and the serial chain is splits / 64 long instead of splits.
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(size_t n, int splits, const float *__restrict__ part, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    float acc = 0.0f;
    for (int s = lane; s < splits; s += 64) acc += part[(size_t)s * n + i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[i] = acc;
}

// dW and db of one weight gradient in ONE launch: blocks 0 .. nb1-1 reduce (n1, part1 -> out1), the rest (n2, part2 -> out2), each with the
// per-output summation order of the kernel above that launch_reduce_partials would have picked for it (results are bit-identical)
__global__ __launch_bounds__(256) void reduce_partials_pair_kernel(size_t n1, size_t n2, int splits, const float *__restrict__ part1, float *__restrict__ out1,
                                                                   const float *__restrict__ part2, float *__restrict__ out2, int nb1, int wide1, int wide2)
{
    const bool second = (int)blockIdx.x >= nb1;
    const size_t n = second ? n2 : n1;
    const float *part = second ? part2 : part1;
    float *out = second ? out2 : out1;
    const unsigned b = second ? blockIdx.x - nb1 : blockIdx.x;
    if (second ? wide2 : wide1) {
        const size_t i = (size_t)b * 4 + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        if (i >= n) return;
        float acc = 0.0f;
        for (int s = lane; s < splits; s += 64) acc += part[(size_t)s * n + i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) out[i] = acc;
    } else {
        const size_t i = (size_t)b * 256 + threadIdx.x;
        if (i < n) {
            float acc = 0.0f;
            for (int s = 0; s < splits; ++s) acc += part[(size_t)s * n + i];
            out[i] = acc;
        }
    }
}
static void launch_reduce_partials_pair(size_t n1, size_t n2, int splits, const float *part1, float *out1, const float *part2, float *out2, hipStream_t st)
{
    const int w1 = splits >= 48 && n1 <= 16384, w2 = splits >= 48 && n2 <= 16384;
    const int nb1 = (int)(w1 ? (n1 + 3) / 4 : (n1 + 255) / 256), nb2 = (int)(w2 ? (n2 + 3) / 4 : (n2 + 255) / 256);
    hipLaunchKernelGGL(reduce_partials_pair_kernel, dim3(nb1 + nb2), dim3(256), 0, st, n1, n2, splits, part1, out1, part2, out2, nb1, w1, w2);
}

static void launch_reduce_partials(size_t n, int splits, const float *part, float *out, hipStream_t st)
{
    if (splits >= 48 && n <= 16384) hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, splits, part, out);
    else hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, splits, part, out);
}

} // namespace
