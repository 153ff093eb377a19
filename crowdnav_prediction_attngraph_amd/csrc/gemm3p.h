// gemm3p.h -- the split-precision ("bf16x3") NT GEMM of gemm3.h for the PPO update's row counts (M = 3e5 .. 5e5 live rows).
//
// Why a second kernel.  Knock-out runs of the 128 x 128 / two-barrier kernel on the q|k|v shape (M = 400 k, N = 1536, K = 512;
// 2.83 ms) showed its four components adding up instead of overlapping: MFMA 0.95 ms + LDS / convert / barrier skeleton 0.81
// + global loads 0.65 + C stores 0.45.  Neither more workgroups per CU, larger tiles nor a double-buffered K loop changed the
// sum (all within 7 %): every wavefront of a workgroup is in the same phase at the same time and a wavefront that is issuing
// MFMAs issues nothing else; and the stores sat one per basic block behind an s_waitcnt vmcnt(0) (see the epilogue).
// This kernel puts the phases into ONE instruction stream per wavefront and takes each operand over the path that suits it:
//   * workgroup tile 128 x (128 NB), four wavefronts side by side: each owns all 128 rows (MI = 4) and 32 NB columns,
//   * A (fp32 activations): coalesced 128-byte row segments -> registers -> hi/lo -> LDS (20 KB per K tile of 32, two buffers),
//     read back as fragments by all four wavefronts; W (bf16 hi/lo, L2 resident): stored by cn_split_bf16 in MFMA FRAGMENT
//     ORDER, so a wavefront's fragment is one contiguous 1 KB load straight into registers -- no LDS, no redundancy,
//   * ONE barrier per K tile; phase A: MFMAs of k-step 0 run over { A fragment reads of k-step 1, conversion + LDS stores of the
//     next A tile, global loads of the tile after it }; phase B: MFMAs of k-step 1 over { A fragment reads of the next tile's
//     k-step 0, W fragment loads two tiles ahead }; groups of three MFMAs fenced with sched_barrier keep the side work spread.
// LDS traffic per K tile: 16 KB written + 64 KB read for 48 MFMAs (1536 cycles) per wavefront = 40 % of the LDS peak.
#pragma once
#include "gemm3.h"

namespace {

// W in fragment order: element (cb, kk, lane, e) = W'[cb * 32 + (lane & 31)][kk * 16 + (lane >> 5) * 8 + e], W' = w or w^T
// Nreal <= Nw: rows Nreal .. Nw-1 of W' are zero padding (a weight whose row count is not a multiple of the 128-column tile)
__global__ void split_bf16_frag_kernel(int Nw, int Kw, int transpose, const float *__restrict__ w, __bf16 *__restrict__ hi, __bf16 *__restrict__ lo, int Nreal)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Nw * Kw) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const size_t blk = idx >> 9;
    const int kk = (int)(blk % (Kw / 16)), cb = (int)(blk / (Kw / 16));
    const int n = cb * 32 + (lane & 31), k = kk * 16 + (lane >> 5) * 8 + e;
    const float x = n >= Nreal ? 0.0f : (transpose ? w[(size_t)k * Nreal + n] : w[(size_t)n * Kw + k]);
    const __bf16 h = (__bf16)x;
    hi[idx] = h;
    lo[idx] = (__bf16)(x - (float)h);
}

// several weights in one launch (the robot-node sequence splits eleven per optimiser step): block b serves the job whose block range holds it
struct SplitGroupJob {
    const float *w;
    __bf16 *hi, *lo;
    int Nw, Kw, transpose, Nreal, block0;
};
struct SplitGroupTable {
    int njobs;
    SplitGroupJob job[20];
};
__global__ void split_bf16_group_kernel(const SplitGroupTable tab)
{
    int ji = 0;
    while (ji + 1 < tab.njobs && (int)blockIdx.x >= tab.job[ji + 1].block0) ++ji;
    const SplitGroupJob &J = tab.job[ji];
    const size_t idx = (size_t)((int)blockIdx.x - J.block0) * blockDim.x + threadIdx.x;
    if (J.Kw == 0) { // a bias vector, zero-padded to Nw floats
        if (idx < (size_t)J.Nw) reinterpret_cast<float *>(J.hi)[idx] = idx < (size_t)J.Nreal ? J.w[idx] : 0.0f;
        return;
    }
    const int Nw = J.Nw, Kw = J.Kw, Nreal = J.Nreal;
    if (idx >= (size_t)Nw * Kw) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const size_t blk = idx >> 9;
    const int kk = (int)(blk % (Kw / 16)), cb = (int)(blk / (Kw / 16));
    const int n = cb * 32 + (lane & 31), k = kk * 16 + (lane >> 5) * 8 + e;
    const float x = n >= Nreal ? 0.0f : (J.transpose ? J.w[(size_t)k * Nreal + n] : J.w[(size_t)n * Kw + k]);
    const __bf16 h = (__bf16)x;
    J.hi[idx] = h;
    J.lo[idx] = (__bf16)(x - (float)h);
}

// KO (experiments only): bit 0 = no C stores, bit 2 = no MFMA, bit 3 = no global loads in the loop
template <int NB, int ACT, bool GATE, int KO = 0>
__global__ __launch_bounds__(256, 1) void gemm3p_nt_kernel(int M, int N, int K, const float *__restrict__ A, int lda, const float *__restrict__ Agate,
                                                           const __bf16 *__restrict__ Whi, const __bf16 *__restrict__ Wlo,
                                                           const float *__restrict__ bias, float *__restrict__ C, int ldc,
                                                           const float *__restrict__ aux, int ldaux, int relu_from)
{
    // Epilogues (ACT, gemm.h): none / ReLU / tanh on the sum, or -- backward products of the robot-node sequence -- the sum times the
    // derivative of the activation whose forward VALUE y sits in aux [M, ldaux]: [y > 0] (ACT_MUL_DRELU) or 1 - y^2 (ACT_MUL_DTANH).
    // Columns >= relu_from get a ReLU on top of ACT (one launch produces [u | relu(enc)]); pass relu_from >= N for none.
    constexpr int TBM = 128, MI = 4, BN = 128 * NB;
    constexpr int PK = 32, PS = 40;     // K tile (two k-steps), LDS row stride in bf16 (80 B: conflict-free 16-byte fragment reads)
    constexpr int BUF = 2 * TBM * PS;   // bf16 elements per LDS buffer: A hi and lo planes of one K tile (20 480 B)
    // PERSISTENT workgroups: one per CU, grid = 8 x slots.  Workgroup L runs on XCD L % 8 (round-robin dispatch) and walks the tiles
    // q = slot, slot + slots, ... of that XCD in the order of xcd_tile / xcd_tile_split (gemm.h: the column tiles of a row tile are
    // consecutive, so the ~32 tiles in flight on an XCD share five or six A row tiles), WITHOUT draining the pipeline in between:
    // the last K tiles of one output tile already request / stage the first K tiles of the next.  Per-tile overhead (two exposed
    // memory round trips of the prologue, the store drain, the dispatch) was 8 us of the 29 us a q|k|v tile takes.
    const int nbx = N / BN, gy = (((M + TBM - 1) / TBM) + 7) & ~7, Qx = nbx * gy / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const bool wide = (size_t)N * K > (size_t)512 * 1024 && !(nbx & 1); // W planes > 2 MB: halve the per-L2 W set (xcd_tile_split)
    auto tile_of = [&](int q, int &row_tile, int &col_tile) {
        if (wide) { const int ch = nbx >> 1; col_tile = (xcd >> 2) * ch + q % ch; row_tile = (q / ch) * 4 + (xcd & 3); }
        else { col_tile = q % nbx; row_tile = (q / nbx) * 8 + xcd; }
    };
    extern __shared__ __attribute__((aligned(16))) char smem3p[];
    __bf16 *lds = reinterpret_cast<__bf16 *>(smem3p);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int T = K / PK, KK = K / 16;
    int q = slot, row_tile, col_tile;
    if (q >= Qx) return;
    tile_of(q, row_tile, col_tile);
    if (row_tile * TBM >= M) return; // (row tiles grow with q: nothing further on either)
    int m_blk = row_tile * TBM, n_blk = col_tile * BN + wave * 32 * NB;
    // (starting the K walk of each workgroup at a different tile -- the workgroups of an XCD stream the same W rows -- changed
    // nothing: 2.10 vs 2.15 ms; L2 channel hot-spotting is not the bound here)

    f32x16 acc[MI][NB];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // A staging: 8 lanes cover one 128-byte row segment, 32 rows per pass, 4 passes
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    int ao[4], ao_n[4];      // element offsets of this thread's four A rows (+ column) in this output tile / the next one (M lda < 2^31)
    int w_base, w_base_n;    // fragment (cb, kk) starts at element ((cb * KK + kk) * 64 + lane) * 8 of a W plane
    int m_blk_n, n_blk_n;
    bool more;
    auto point_at = [&](int mb, int nb, int (&a)[4], int &wb) {
#pragma unroll
        for (int p = 0; p < 4; ++p) a[p] = min(mb + arow + 32 * p, M - 1) * lda + acol; // surplus rows repeat row M-1: computed, never stored
        wb = ((nb >> 5) * KK * 64 + lane) * 8;
    };
    auto look_ahead = [&]() { // the tile after (m_blk, n_blk); without one the "next" tile is this one again (its loads are never used)
        int rt, ct;
        more = q + slots < Qx;
        if (more) { tile_of(q + slots, rt, ct); more = rt * TBM < M; }
        m_blk_n = more ? rt * TBM : m_blk;
        n_blk_n = more ? ct * BN + wave * 32 * NB : n_blk;
        point_at(m_blk_n, n_blk_n, ao_n, w_base_n);
    };
    point_at(m_blk, n_blk, ao, w_base);
    look_ahead();
    // W fragment sets / staged A tiles in flight (prefetch depth in K tiles).  4 / 2 measured the same as 2 / 1 (2.16 ms on the
    // q|k|v shape either way): the loop is not waiting for latency, so the smaller register footprint stays
    constexpr int WD = 2, AD = 1;
    f32x4 sa[AD][4], sg[GATE ? AD : 1][4];
    bf16x8 fah[2][MI], fal[2][MI];   // [k-step][block] hi / lo fragments of A
    bf16x8 fwh[WD][2][NB], fwl[WD][2][NB]; // [tile % WD][k-step][block] hi / lo fragments of W
    // K tile indices T, T + 1, ... are K tiles 0, 1, ... of the next output tile
    auto load_a = [&](int set, int p, int tile) {
        if (KO & 8) return;
        const bool nx = tile >= T;
        const size_t o = (size_t)((nx ? ao_n[p] : ao[p]) + (nx ? tile - T : tile) * PK);
        sa[set][p] = *reinterpret_cast<const f32x4 *>(A + o);
        if (GATE) sg[GATE ? set : 0][p] = *reinterpret_cast<const f32x4 *>(Agate + o); // the select happens at the conversion
    };
    auto stage_a = [&](int set, int p, int b) { // convert pass p of a staged tile into buffer b
        __bf16 *Ah = lds + b * BUF, *Al = Ah + TBM * PS;
        bf16x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = GATE ? (sg[GATE ? set : 0][p][q] > 0.0f ? sa[set][p][q] : 0.0f) : sa[set][p][q];
            hi[q] = (__bf16)a;
            lo[q] = (__bf16)(a - (float)hi[q]);
        }
        *reinterpret_cast<bf16x4 *>(&Ah[(arow + 32 * p) * PS + acol]) = hi;
        *reinterpret_cast<bf16x4 *>(&Al[(arow + 32 * p) * PS + acol]) = lo;
    };
    const int a_off = l31 * PS + half * 8;
    auto read_a = [&](int ks, int s, int b) { // one of the 8 A fragment reads of a k-step
        const __bf16 *Ah = lds + b * BUF + ks * 16 + a_off, *Al = Ah + TBM * PS;
        if (s < 4) fah[ks][s] = *reinterpret_cast<const bf16x8 *>(&Ah[s * 32 * PS]);
        else fal[ks][s - 4] = *reinterpret_cast<const bf16x8 *>(&Al[(s - 4) * 32 * PS]);
    };
    auto load_w = [&](int par, int ks, int j, int tile) {
        if (KO & 8) return;
        const bool nx = tile >= T;
        const size_t o = (size_t)((nx ? w_base_n : w_base) + (j * KK + (nx ? tile - T : tile) * 2 + ks) * 512);
        fwh[par][ks][j] = *reinterpret_cast<const bf16x8 *>(Whi + o);
        fwl[par][ks][j] = *reinterpret_cast<const bf16x8 *>(Wlo + o);
    };
    // one MFMA of a k-step: term-major order (lo*hi, hi*lo, hi*hi; consecutive MFMAs hit different accumulators)
    auto mfma_one = [&](int par, int ks, int s) {
        const int t = s / (MI * NB), i = (s % (MI * NB)) / NB, j = s % NB;
        if (KO & 4) { acc[i][j][0] += (float)fal[ks][i][t] * (float)fwh[par][ks][j][0] + (float)fah[ks][i][1] * (float)fwl[par][ks][j][t]; return; }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 0 ? fal[ks][i] : fah[ks][i], t == 1 ? fwl[par][ks][j] : fwh[par][ks][j], acc[i][j], 0, 0, 0);
    };
    constexpr int G = MI * NB, RPG = 8 / G, SPG = G / 4; // groups of three MFMAs per k-step; A fragment reads per group; groups per staging pass

    // prologue: A tile 0 in LDS buffer 0, tiles 1 .. AD staged in registers; W fragments of tiles 0 .. WD-1 requested
#pragma unroll
    for (int p = 0; p < 4; ++p) load_a(0, p, 0);
#pragma unroll
    for (int u = 0; u < WD; ++u)
#pragma unroll
        for (int j = 0; j < NB; ++j) { load_w(u, 0, j, u); load_w(u, 1, j, u); }
    if (AD > 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) load_a(1 % AD, p, 1);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) { stage_a(0, p, 0); load_a(0, p, AD); }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) read_a(0, s, 0);
    float bv[NB]; // this output tile's bias (requested a whole tile ahead of its use)
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[j] = bias ? bias[n_blk + j * 32 + l31] : 0.0f;
    for (;;) {
    // T % WD == 0 (K % 64 == 0): WD tiles per trip so that register sets are compile-time indices
    for (int t = 0; t < T; t += WD) {
#pragma unroll
        for (int u = 0; u < WD; ++u) {
            const int cur = u & 1;      // tile t + u lives in LDS buffer (t + u) & 1
            const int nset = (u + 1) % AD; // the staged copy of tile t + u + 1
            // Branch-free body.  Fenced groups of three MFMAs keep the side work spread out under them.
            const int ta = t + u + 1 + AD, tw = t + u + WD;
            // ---- phase A: k-step 0 | A fragments of k-step 1; the staged A tile (t+u+1) converted into the other buffer and the
            //      tile AD further on requested ----
#pragma unroll
            for (int g = 0; g < G; ++g) {
                mfma_one(u, 0, 3 * g);
#pragma unroll
                for (int r = 0; r < RPG; ++r) read_a(1, g * RPG + r, cur);
                mfma_one(u, 0, 3 * g + 1);
                if (g % SPG == 0) { stage_a(nset, g / SPG, cur ^ 1); load_a(nset, g / SPG, ta); }
                mfma_one(u, 0, 3 * g + 2);
                // the k-step-1 W registers of the PREVIOUS tile's set are free since its phase B: the tile WD further on goes
                // there, one fragment per group (in the very first trip this re-requests what the prologue put there)
                if (g >= G - NB) load_w((u + WD - 1) % WD, 1, g - (G - NB), t + u - 1 + WD);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads(); // A tile t+u+1 is complete, the buffer of tile t+u is free
            // ---- phase B: k-step 1 | A fragments (k-step 0) of tile t+u+1 ----
#pragma unroll
            for (int g = 0; g < G; ++g) {
                mfma_one(u, 1, 3 * g);
#pragma unroll
                for (int r = 0; r < RPG; ++r) read_a(0, g * RPG + r, cur ^ 1);
                mfma_one(u, 1, 3 * g + 1);
                if (g < NB) load_w(u, 0, g, tw); // this set's k-step-0 registers are free since phase A
                mfma_one(u, 1, 3 * g + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // Epilogue of this output tile (the pipeline already holds the first K tiles of the next one).  A workgroup whose 128 rows are
    // all inside M stores without per-row predicates: with the predicate, every store sat in its own basic block behind an
    // s_waitcnt vmcnt(0) (the compiler cannot tell there that the bias load has landed), i.e. every store waited for the previous
    // one to complete.
    auto store_all = [&](auto guard) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int row0 = m_blk + i * 32 + 4 * half, col = n_blk + j * 32 + l31;
                float *cp = C + (size_t)row0 * ldc + col;
                constexpr bool AUX = ACT == ACT_MUL_DRELU || ACT == ACT_MUL_DTANH;
                float y[AUX ? 16 : 1]; // the block's 16 activation values first, then the 16 stores (a load between two stores would wait for the first)
                if (AUX) {
                    const float *ap = aux + (size_t)row0 * ldaux + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        y[AUX ? r : 0] = guard(row0 + ro) ? ap[(size_t)ro * ldaux] : 0.0f;
                    }
                }
                const bool extra_relu = col >= relu_from;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r] + bv[j];
                    if (ACT == ACT_RELU || extra_relu) v = fmaxf(v, 0.0f);
                    if (ACT == ACT_TANH) v = tanhf(v);
                    if (ACT == ACT_MUL_DRELU) v = y[AUX ? r : 0] > 0.0f ? v : 0.0f;
                    if (ACT == ACT_MUL_DTANH) v *= 1.0f - y[AUX ? r : 0] * y[AUX ? r : 0];
                    acc[i][j][r] = 0.0f;
                    if ((KO & 1) && v != 12345.678f) continue;
                    if (guard(row0 + ro)) __builtin_nontemporal_store(v, cp + (size_t)ro * ldc); // streaming result: keep A / W in the L2
                }
            }
    };
    if (m_blk + TBM <= M) store_all([](int) { return true; });
    else store_all([&](int row) { return row < M; });
    if (!more) break;
    // the next output tile becomes the current one
    q += slots;
    m_blk = m_blk_n; n_blk = n_blk_n; w_base = w_base_n;
#pragma unroll
    for (int p = 0; p < 4; ++p) ao[p] = ao_n[p];
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[j] = bias ? bias[n_blk + j * 32 + l31] : 0.0f;
    look_ahead();
    }
}

template <int ACT, int KO = 0>
static int launch_gemm3p(int M, int N, int K, const float *A, int lda, const __bf16 *Whi, const __bf16 *Wlo, const float *bias, float *C, int ldc,
                         hipStream_t st, const float *Agate, const float *aux = nullptr, int ldaux = 0, int relu_from = 1 << 30)
{
    CN_REQUIRE(N % 128 == 0 && K % 64 == 0 && lda % 4 == 0, "gemm3p: unsupported shape M=%d N=%d K=%d lda=%d", M, N, K, lda);
    CN_REQUIRE((long long)M * lda < (1LL << 31) && (long long)N * K < (1LL << 31), "gemm3p: operand too large for 32-bit element offsets (M=%d lda=%d)", M, lda);
    if (M == 0) return CN_OK;
    constexpr size_t lds = (size_t)2 * 2 * 128 * 40 * sizeof(__bf16); // 40 960 B
    const int gy = (((M + 127) / 128) + 7) & ~7;
    const int nb = N % 256 == 0 ? 2 : 1;
    const int per_xcd = N / (128 * nb) * gy / 8;                 // output tiles per XCD
    const dim3 grid(8 * (per_xcd < 32 ? per_xcd : 32));          // persistent: one workgroup per CU, 32 CUs per XCD
    if (nb == 2) {
        if (Agate) hipLaunchKernelGGL((gemm3p_nt_kernel<2, ACT, true, KO>), grid, dim3(256), lds, st, M, N, K, A, lda, Agate, Whi, Wlo, bias, C, ldc, aux, ldaux, relu_from);
        else hipLaunchKernelGGL((gemm3p_nt_kernel<2, ACT, false, KO>), grid, dim3(256), lds, st, M, N, K, A, lda, Agate, Whi, Wlo, bias, C, ldc, aux, ldaux, relu_from);
    } else {
        if (Agate) hipLaunchKernelGGL((gemm3p_nt_kernel<1, ACT, true, KO>), grid, dim3(256), lds, st, M, N, K, A, lda, Agate, Whi, Wlo, bias, C, ldc, aux, ldaux, relu_from);
        else hipLaunchKernelGGL((gemm3p_nt_kernel<1, ACT, false, KO>), grid, dim3(256), lds, st, M, N, K, A, lda, Agate, Whi, Wlo, bias, C, ldc, aux, ldaux, relu_from);
    }
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ---- weight gradient: P[s][n][k] = sum_{m in split s} dY[m][n] X[m][k] (see gemm3_tn_kernel in gemm3.h for the contract) ----
// Same division of labour as gemm3p_nt_kernel: the 128 dY columns of the tile are shared by the four wavefronts and go through
// the LDS (transposed on the way in: a thread loads 8 consecutive m of ONE column with 8 coalesced dword loads and stores them as
// one 16-byte fragment word per plane), the X columns belong to exactly one wavefront each (32 NB of them) and go straight from
// global memory into fragment registers: lane (l31, half) of block j loads X[m0 + 8 half + e][k0 + 32 j + l31], e = 0..7 -- eight
// dword loads of two full 128-byte lines each.  One barrier per 32 rows of m; no LDS traffic for X at all.  NB = 1, 2 or 4 (a wavefront
// then owns 32 / 64 / 128 X columns and 64 / 128 / 256 accumulator registers).
template <int NB, bool GATE>
__global__ __launch_bounds__(256, 1) void gemm3p_tn_kernel(int M, int N, int K, const float *__restrict__ dY, int ldy, const float *__restrict__ Ygate,
                                                           const float *__restrict__ X, int ldx, int rows_per_split, int nsplit,
                                                           float *__restrict__ partials, float *__restrict__ db_part)
{
    constexpr int MI = 4, PS = 40, BUF = 2 * 128 * PS;
    extern __shared__ __attribute__((aligned(16))) char smem3p[];
    __bf16 *lds = reinterpret_cast<__bf16 *>(smem3p);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (uniform: the buffer resources below live in SGPRs)
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware 1-D grid: workgroup L runs on XCD L % 8 (round-robin dispatch); all tiles of one split go to ONE XCD, so the rows
    // of dY and X that the split owns stream through one L2 once (the tiles advance over m together) instead of through up to 8 of
    // them (with x-fastest 3-D indexing every tile of a split sat on a different XCD: 14.7 GB of L2 misses for 3.3 GB of operands)
    const int L = blockIdx.x, NX = N / 128, tiles = NX * (K / (128 * NB));
    const int split = (L & 7) + 8 * ((L >> 3) / tiles), tile = (L >> 3) % tiles;
    if (split >= nsplit) return;
    const int n_blk = (tile % NX) * 128, k_blk = (tile / NX) * (128 * NB) + wave * 32 * NB;
    // M and rows_per_split are multiples of 32 here (the launcher hands the last M % 32 rows to gemm3_tn_kernel): no row predicates,
    // and every row offset below is wave-uniform, i.e. scalar address arithmetic (one SALU add per load instead of a 64-bit VALU chain)
    const int m_begin = split * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    const int T = (m_end - m_begin) / 32;
    const bool want_db = db_part != nullptr && tile / NX == 0;

    f32x16 acc[MI][NB];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // dY staging: thread (column c, half-tile g0) owns m groups g0 and g0 + 2 (8 rows each) of the 32-row tile
    const int c = tid & 127, g0 = __builtin_amdgcn_readfirstlane(tid >> 7);
    // Buffer addressing relative to the split's first row: a load is ONE instruction (per-lane byte offset register + wave-uniform
    // scalar offset + immediate).  With flat 64-bit addresses every one of the 48 loads of a 32-row tile carried a 64-bit VALU add
    // and three scalar multiplies / adds -- 130 of the loop's 460 instructions, on a wavefront that is alone on its SIMD.
    // (a split spans < 2^31 bytes of either operand: rows_per_split * ld * 4)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)(dY + (size_t)m_begin * ldy + n_blk), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)((GATE ? Ygate : dY) + (size_t)m_begin * ldy + n_blk), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (size_t)m_begin * ldx + k_blk), 0, 0x7fffffff, 0x00020000);
    unsigned yv[8], xv[8]; // per-lane byte offsets of row e of a group: column c of dY; the lane's X column, its 8 rows start 8 * half below the k-step's first row
#pragma unroll
    for (int e = 0; e < 8; ++e) { yv[e] = (unsigned)(e * ldy + c) * 4u; xv[e] = (unsigned)((half * 8 + e) * ldx + l31) * 4u; }
    float sy[2][8], sg[GATE ? 2 : 1][8];
    // X: raw rows of this lane's fragments, [k-step][block][e]
    float rx[2][NB][8];
    bf16x8 fah[2][MI], fal[2][MI], fwh[2][NB], fwl[2][NB];
    float colsum = 0.0f;
    // tiles past the end of the split (the pipeline runs two ahead) read its last tile again; what they stage is never multiplied
    auto load_y = [&](int q, int tile) {
        const unsigned so = (unsigned)((min(tile, T - 1) * 32 + (g0 + 2 * q) * 8) * ldy) * 4u; // wave-uniform
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sy[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, yv[e], so, 0));
            if (GATE) sg[GATE ? q : 0][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, yv[e], so, 0));
        }
    };
    auto stage_y = [&](int q, int tile, int b) {
        const float cm = tile < T ? 1.0f : 0.0f; // the column sums count every row once
        __bf16 *Ah = lds + b * BUF, *Al = Ah + 128 * PS;
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = sy[q][e];
            if (GATE) a = sg[GATE ? q : 0][e] > 0.0f ? a : 0.0f;
            colsum = fmaf(cm, a, colsum);
            const __bf16 h = (__bf16)a;
            hi[e] = h;
            lo[e] = (__bf16)(a - (float)h);
        }
        *reinterpret_cast<bf16x8 *>(&Ah[c * PS + (g0 + 2 * q) * 8]) = hi;
        *reinterpret_cast<bf16x8 *>(&Al[c * PS + (g0 + 2 * q) * 8]) = lo;
    };
    auto load_x = [&](int ks, int j, int tile) {
        const unsigned so = (unsigned)((min(tile, T - 1) * 32 + ks * 16) * ldx) * 4u; // wave-uniform
#pragma unroll
        for (int e = 0; e < 8; ++e) rx[ks][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xv[e] + 128u * j, so, 0));
    };
    auto convert_x = [&](int ks, int j) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = rx[ks][j][e];
            const __bf16 h = (__bf16)a;
            fwh[ks][j][e] = h;
            fwl[ks][j][e] = (__bf16)(a - (float)h);
        }
    };
    const int a_off = l31 * PS + half * 8;
    auto read_a = [&](int ks, int s, int b) {
        const __bf16 *Ah = lds + b * BUF + ks * 16 + a_off, *Al = Ah + 128 * PS;
        if (s < 4) fah[ks][s] = *reinterpret_cast<const bf16x8 *>(&Ah[s * 32 * PS]);
        else fal[ks][s - 4] = *reinterpret_cast<const bf16x8 *>(&Al[(s - 4) * 32 * PS]);
    };
    auto mfma_one = [&](int ks, int s) {
        const int t = s / (MI * NB), i = (s % (MI * NB)) / NB, j = s % NB;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 0 ? fal[ks][i] : fah[ks][i], t == 1 ? fwl[ks][j] : fwh[ks][j], acc[i][j], 0, 0, 0);
    };
    constexpr int G = MI * NB, RPG = 8 / G; // groups of three MFMAs per k-step; dY fragment reads per group

    // prologue: dY tile 0 in LDS buffer 0, tile 1 staged; X tile 0 converted (k-step 0) / raw (k-step 1), tile 1 k-step 0 requested
    load_y(0, 0); load_y(1, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) { load_x(0, j, 0); load_x(1, j, 0); }
    stage_y(0, 0, 0); stage_y(1, 0, 0);
    load_y(0, 1); load_y(1, 1);
#pragma unroll
    for (int j = 0; j < NB; ++j) { convert_x(0, j); load_x(0, j, 1); }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) read_a(0, s, 0);
    int cur = 0;
    for (int t = 0; t < T; ++t) {
        // Branch-free body.
        // ---- phase A: k-step 0 | dY fragments of k-step 1; dY tile t+1 into the other buffer, tile t+2 requested; X k-step 1 of
        //      tile t converted, of tile t+1 requested ----
#pragma unroll
        for (int g = 0; g < G; ++g) {
            mfma_one(0, 3 * g);
#pragma unroll
            for (int r = 0; r < RPG; ++r) read_a(1, g * RPG + r, cur);
            if (RPG == 0 && g % (G / 8) == 0) read_a(1, g / (G / 8), cur);
            mfma_one(0, 3 * g + 1);
            if (g == 0 || g == G / 2) { const int q = g ? 1 : 0; stage_y(q, t + 1, cur ^ 1); load_y(q, t + 2); }
            if (g % (G / NB) == G / NB - 1) { const int j = g / (G / NB); convert_x(1, j); load_x(1, j, t + 1); }
            mfma_one(0, 3 * g + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads(); // dY tile t+1 is complete, the buffer of tile t is free
        // ---- phase B: k-step 1 | dY fragments (k-step 0) of tile t+1; X k-step 0 of tile t+1 converted, of tile t+2 requested ----
#pragma unroll
        for (int g = 0; g < G; ++g) {
            mfma_one(1, 3 * g);
#pragma unroll
            for (int r = 0; r < RPG; ++r) read_a(0, g * RPG + r, cur ^ 1);
            if (RPG == 0 && g % (G / 8) == 0) read_a(0, g / (G / 8), cur ^ 1);
            mfma_one(1, 3 * g + 1);
            if (g % (G / NB) == G / NB - 1) { const int j = g / (G / NB); convert_x(0, j); load_x(0, j, t + 2); }
            mfma_one(1, 3 * g + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= 1;
    }
    float *P = partials + (size_t)split * N * K;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float *pp = P + (size_t)(n_blk + i * 32 + 4 * half) * K + k_blk + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) pp[(size_t)((r & 3) + 8 * (r >> 2)) * K] = acc[i][j][r];
        }
    if (want_db) { // uniform per block
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem3p);
        red[tid] = colsum;
        __syncthreads();
        if (tid < 128) db_part[(size_t)split * N + n_blk + tid] = red[tid] + red[tid + 128];
    }
}

} // namespace
