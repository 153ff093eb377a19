// linear.hip -- the large Linear layers of the PPO update (forward, input gradient, weight/bias gradient) on gfx950 in
// the same split-precision bf16x3 MFMA arithmetic as the rollout forward (gemm3.h).
// Reference: the torch.nn.Linear calls of SpatialEdgeSelfAttn.forward / spatial_linear
// (rl/networks/selfAttn_srnn_temp_node.py:63-91,408) executed under autograd by PPO.update (rl/ppo.py:60-95:
// evaluate_actions -> loss.backward()).
#include "common.h"
#include "gemm3p.h"
#include "train_internal.h"

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
    // eight rows per trip, all their loads requested before the first use (one row per trip = one memory round trip per row)
    for (int r0 = blockIdx.x; r0 < R; r0 += 8 * gridDim.x) {
        float yv[8], dv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = min(r0 + k * (int)gridDim.x, R - 1);
            yv[k] = y[(size_t)r * 128 + n];
            dv[k] = dy[(size_t)r * 128 + n];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = r0 + k * (int)gridDim.x;
            const float g = (r < R && yv[k] > 0.0f) ? dv[k] : 0.0f;
            const float *xr = x + (size_t)min(r, R - 1) * D;
#pragma unroll
            for (int d = 0; d < 16; ++d)
                if (d < D) acc[d] += g * xr[d];
            acc[16] += g;
        }
    }
    float *p = part + ((size_t)blockIdx.x * 128 + n) * (D + 1); // [block][n][D weights | bias]
#pragma unroll
    for (int d = 0; d < 16; ++d)
        if (d < D) p[d] = acc[d];
    p[D] = acc[16];
}


// ---- whole GRU sequence in one launch (forward) / one launch (backward) ---------------------------------------------------
// The update runs the human-node GRU over T = 30 steps with only N x 128 of state: as separate launches (a BLAS product and
// a pointwise kernel per step and direction) it is launch- and latency-bound.  Here a workgroup owns 16 rows (envs) for
// the whole sequence (2048 envs of a minibatch = 128 workgroups; with 32 rows it was 64 of 256 CUs) and W_hh never leaves
// the register file:
//   * forward: wavefront w owns hidden units 32w .. 32w+31 as two blocks of 16, i.e. six 16-column blocks (gate j, block a);
//     its 96 x 128 slice of W_hh is loaded once and kept as bf16 hi / lo MFMA fragments (192 VGPRs).  Per step the masked state
//     hm (16 x 128, LDS, double buffered: fp32 for the cell's z * h term, bf16 hi / lo planes written by the lane that computed
//     the value) is the A operand of 72 split-precision MFMAs (v_mfma_f32_16x16x32_bf16: lo*hi, hi*lo, hi*hi per column block
//     and k-step); the accumulators of the r, z, n gates of one (row, unit) sit in the same lane and register index, so the
//     cell's pointwise part -- eight cells per lane -- needs no exchange at all; one barrier per step.
//   * backward (reverse time): the same wavefront computes d(gates) for its units from the saved (r, z, n, gh_n), publishes
//     d(gh) [16 x 384] as hi / lo planes in LDS, and multiplies it with its 384 x 32 slice of W_hh (again 192 resident VGPRs) to
//     get d(hm); the carried gradient stays in registers in the accumulator layout.  d(W_hh) is one product over all T*N rows afterwards.
constexpr int GR = 16, GHS = 132;   // rows per workgroup; LDS row stride (floats) of the fp32 state
constexpr int GPS = 136, GQS = 392; // row strides (bf16) of the hi / lo planes of hm / d(gh): 272 / 784 bytes = 4 banks mod 64, so the
                                    // 16-byte fragment reads of 16 rows tile all banks
typedef float f32x4g __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &hi, bf16x8 &lo)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) { const __bf16 h = (__bf16)x[e]; hi[e] = h; lo[e] = (__bf16)(x[e] - (float)h); }
}

// Accumulator layout of the 16 x 16 MFMA: lane (l15 = lane & 15, kg = lane >> 4) holds rows 4 kg + r (r = 0..3) of column l15.
// Operand layout of v_mfma_f32_16x16x32_bf16: lane (l15, kg) supplies 8 consecutive k = 32 kk + 8 kg + e of row / column l15.
__global__ __launch_bounds__(256) void gru_seq_fwd_kernel(int T, int N, const float *__restrict__ gi, const float *__restrict__ h0,
                                                          const float *__restrict__ m, const float *__restrict__ Whh, const float *__restrict__ bhh,
                                                          float *__restrict__ hs, float *__restrict__ hms, float *__restrict__ gates)
{
    __shared__ __attribute__((aligned(16))) float hm[2][GR * GHS];
    __shared__ __attribute__((aligned(16))) __bf16 hmh[2][GR * GPS], hml[2][GR * GPS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4, l15 = lane & 15;
    const int row0 = blockIdx.x * GR;
    // B fragments: column = unit u(a) = 32 wave + 16 a + l15 of gate j, k = 32 kk + 8 kg + e
    bf16x8 wh[3][2][4], wl[3][2][4];
    float bias[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int u = 32 * wave + 16 * a + l15;
            const float *wr = Whh + (size_t)(j * 128 + u) * 128 + 8 * kg;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = wr[32 * kk + e];
                split8(x, wh[j][a][kk], wl[j][a][kk]);
            }
            bias[j][a] = bhh[j * 128 + u];
        }
    for (int i = tid; i < GR * 128; i += 256) {
        const int r = i >> 7, c = i & 127, row = row0 + r;
        const float v = row < N ? h0[(size_t)row * 128 + c] * m[row] : 0.0f;
        const __bf16 h = (__bf16)v;
        hm[0][r * GHS + c] = v; hmh[0][r * GPS + c] = h; hml[0][r * GPS + c] = (__bf16)(v - (float)h);
        if (row < N) hms[(size_t)row * 128 + c] = v;
    }
    __syncthreads();
    // A step's inputs (gi of the step, the mask of the next) do not depend on the state: cell (a, r)'s are requested one step ahead,
    // right after the cell has used the current ones, unpredicated (surplus rows of the last workgroup read row N-1) -- a whole step
    // hides the round trip.  Loaded inside the per-row `if (row < N)` blocks below, every row paid its own memory round trip.
    float gir[2][4], giz[2][4], gin[2][4], mk[4];
    auto request = [&](int a, int r, int t) {
        const int rowc = min(row0 + 4 * kg + r, N - 1);
        const float *gip = gi + ((size_t)min(t, T - 1) * N + rowc) * 384 + 32 * wave + 16 * a + l15;
        gir[a][r] = gip[0]; giz[a][r] = gip[128]; gin[a][r] = gip[256];
        if (a == 0) mk[r] = m[(size_t)min(t + 1, T - 1) * N + rowc];
    };
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) request(a, r, 0);
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        f32x4g acc[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[j][a] = f32x4g{0.f, 0.f, 0.f, 0.f};
        // term-major inside a k-step: consecutive MFMAs write different accumulators
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&hmh[cur][l15 * GPS + 32 * kk + 8 * kg]);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&hml[cur][l15 * GPS + 32 * kk + 8 * kg]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[j][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[j][a][kk], acc[j][a], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[j][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[j][a][kk], acc[j][a], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[j][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[j][a][kk], acc[j][a], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * kg + r, row = row0 + rl, u = 32 * wave + 16 * a + l15;
                const float hn = acc[2][a][r] + bias[2][a];
                const float rg = sigmoidf_(gir[a][r] + acc[0][a][r] + bias[0][a]);
                const float zg = sigmoidf_(giz[a][r] + acc[1][a][r] + bias[1][a]);
                const float ng = tanhf(gin[a][r] + rg * hn);
                const float hnew = (1.0f - zg) * ng + zg * hm[cur][rl * GHS + u];
                const float next = row < N ? hnew * mk[r] : 0.0f;
                if (row < N) { // stores only: nothing in here waits for memory
                    const size_t tr = (size_t)t * N + row;
                    hs[tr * 128 + u] = hnew;
                    float *gp = gates + tr * 512;
                    gp[u] = rg; gp[128 + u] = zg; gp[256 + u] = ng; gp[384 + u] = hn;
                    if (t + 1 < T) hms[(tr + N) * 128 + u] = next;
                }
                if (t + 1 < T) {
                    const __bf16 h = (__bf16)next;
                    hm[cur ^ 1][rl * GHS + u] = next; hmh[cur ^ 1][rl * GPS + u] = h; hml[cur ^ 1][rl * GPS + u] = (__bf16)(next - (float)h);
                }
                if (a == 1) request(0, r, t + 1); // (mk[r] is shared by the two blocks: its reload goes behind the second one's use)
                if (a == 1) request(1, r, t + 1);
            }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gru_seq_bwd_kernel(int T, int N, const float *__restrict__ gates, const float *__restrict__ hms,
                                                          const float *__restrict__ m, const float *__restrict__ Whh, const float *__restrict__ d_hs,
                                                          float *__restrict__ dgi, float *__restrict__ dgh, float *__restrict__ dh0)
{
    __shared__ __attribute__((aligned(16))) __bf16 dgp_h[GR * GQS], dgp_l[GR * GQS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4, l15 = lane & 15;
    const int row0 = blockIdx.x * GR;
    // d(hm)[row][n] = sum_c d(gh)[row][c] * W_hh[c][n]: this lane's columns n = u(a), k = c = 32 kk + 8 kg + e runs over the 384 gate columns
    bf16x8 wh[2][12], wl[2][12];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int kk = 0; kk < 12; ++kk) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = Whh[(size_t)(32 * kk + 8 * kg + e) * 128 + 32 * wave + 16 * a + l15];
            split8(x, wh[a][kk], wl[a][kk]);
        }
    float carry[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) carry[a][r] = 0.0f;
    // The saved gates / states / incoming gradients of a step do not depend on the carried gradient: they are requested one step
    // ahead, unpredicated (surplus rows read row N-1), right behind the barrier that ends their last use -- the MFMAs of the
    // current step hide the round trip.
    float s_rg[2][4], s_zg[2][4], s_ng[2][4], s_hn[2][4], s_h[2][4], s_d[2][4], s_m[4];
    auto preload = [&](int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rowc = min(row0 + 4 * kg + r, N - 1);
            const size_t tr = (size_t)t * N + rowc;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int u = 32 * wave + 16 * a + l15;
                const float *gp = gates + tr * 512 + u;
                s_rg[a][r] = gp[0]; s_zg[a][r] = gp[128]; s_ng[a][r] = gp[256]; s_hn[a][r] = gp[384];
                s_h[a][r] = hms[tr * 128 + u];
                s_d[a][r] = d_hs[tr * 128 + u];
            }
            s_m[r] = m[tr];
        }
    };
    auto put = [&](int off, float v) { const __bf16 h = (__bf16)v; dgp_h[off] = h; dgp_l[off] = (__bf16)(v - (float)h); };
    preload(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        float direct[2][4], mt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = 4 * kg + r, row = row0 + rl;
            const bool ok = row < N;
            mt[r] = s_m[r];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int u = 32 * wave + 16 * a + l15;
                const float rg = s_rg[a][r], zg = s_zg[a][r], ng = s_ng[a][r], hn = s_hn[a][r], h = s_h[a][r];
                const float d = s_d[a][r] + carry[a][r];
                const float din = d * (1.0f - zg) * (1.0f - ng * ng);
                const float dr = ok ? din * hn * rg * (1.0f - rg) : 0.0f;
                const float dz = ok ? d * (h - ng) * zg * (1.0f - zg) : 0.0f;
                const float dn = ok ? din * rg : 0.0f;
                direct[a][r] = ok ? d * zg : 0.0f;
                if (ok) { // stores only
                    const size_t tr = (size_t)t * N + row;
                    float *p1 = dgi + tr * 384, *p2 = dgh + tr * 384;
                    p1[u] = dr; p1[128 + u] = dz; p1[256 + u] = din;
                    p2[u] = dr; p2[128 + u] = dz; p2[256 + u] = dn;
                }
                put(rl * GQS + u, dr); put(rl * GQS + 128 + u, dz); put(rl * GQS + 256 + u, dn);
            }
        }
        __syncthreads();
        preload(t > 0 ? t - 1 : 0);
        // one accumulator per term and column block: 72 MFMAs into two accumulators would each wait for the one before
        f32x4g acc[3][2];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[q][a] = f32x4g{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 12; ++kk) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&dgp_h[l15 * GQS + 32 * kk + 8 * kg]);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&dgp_l[l15 * GQS + 32 * kk + 8 * kg]);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                acc[0][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[a][kk], acc[0][a], 0, 0, 0);
                acc[1][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[a][kk], acc[1][a], 0, 0, 0);
                acc[2][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[a][kk], acc[2][a], 0, 0, 0);
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * kg + r;
                carry[a][r] = row < N ? (direct[a][r] + ((acc[0][a][r] + acc[1][a][r]) + acc[2][a][r])) * mt[r] : 0.0f;
            }
        __syncthreads(); // the next (earlier) step overwrites the planes
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * kg + r;
            if (row < N) dh0[(size_t)row * 128 + 32 * wave + 16 * a + l15] = carry[a][r];
        }
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
    launch_reduce_partials(n, blocks, partials, dWb, st);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_gru_seq_fwd(int T, int N, const float *gi, const float *h0, const float *masks, const float *w_hh, const float *b_hh, float *hs,
                              float *hms, float *gates, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(T >= 1 && N >= 1 && gi && h0 && masks && w_hh && b_hh && hs && hms && gates, "cn_gru_seq_fwd: bad argument");
    hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3((N + GR - 1) / GR), dim3(256), 0, (hipStream_t)stream, T, N, gi, h0, masks, w_hh, b_hh, hs, hms, gates);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_gru_seq_bwd(int T, int N, const float *gates, const float *hms, const float *masks, const float *w_hh, const float *d_hs, float *dgi,
                              float *dgh, float *dh0, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(T >= 1 && N >= 1 && gates && hms && masks && w_hh && d_hs && dgi && dgh && dh0, "cn_gru_seq_bwd: bad argument");
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3((N + GR - 1) / GR), dim3(256), 0, (hipStream_t)stream, T, N, gates, hms, masks, w_hh, d_hs, dgi, dgh, dh0);
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

extern "C" int cn_split_bf16_padded(const float *w, int rows, int cols, int transpose, int n_padded, void *hi, void *lo, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(w && hi && lo && rows > 0 && cols > 0, "cn_split_bf16: bad argument");
    hipStream_t st = (hipStream_t)stream;
    // W' = w (transpose = 0) or w^T: [Nreal, Kw], zero-padded to n_padded rows; stored in the MFMA fragment order gemm3p_nt_kernel loads (gemm3p.h)
    const int Nreal = transpose ? cols : rows, Kw = transpose ? rows : cols;
    const int Nw = n_padded > 0 ? n_padded : Nreal;
    CN_REQUIRE(Nw >= Nreal && Nw % 32 == 0 && Kw % 16 == 0, "cn_split_bf16: the weight seen by the product must be [32 a, 16 b], got [%d (padded %d), %d]", Nreal, Nw, Kw);
    const size_t n = (size_t)Nw * Kw;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(split_bf16_frag_kernel, dim3(blocks), dim3(256), 0, st, Nw, Kw, transpose, w, (__bf16 *)hi, (__bf16 *)lo, Nreal);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
CnSplitJob cn_split_job(const float *w, int rows, int cols, int transpose, int n_padded, float *planes)
{
    const int Nreal = transpose ? cols : rows, Kw = transpose ? rows : cols, Nw = n_padded > 0 ? n_padded : Nreal;
    uint16_t *hi = reinterpret_cast<uint16_t *>(planes);
    return CnSplitJob{w, hi, hi + (size_t)Nw * Kw, Nw, Kw, transpose, Nreal, 0};
}

int cn_split_group_launch(CnSplitJob *jobs, int n, hipStream_t st)
{
    CN_REQUIRE(n >= 0 && n <= CN_SPLIT_MAX_JOBS, "cn_split_group_launch: %d jobs (at most %d)", n, CN_SPLIT_MAX_JOBS);
    if (n == 0) return CN_OK;
    SplitGroupTable tab{};
    tab.njobs = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const CnSplitJob &j = jobs[i];
        CN_REQUIRE(j.w && j.hi && j.Nw >= j.Nreal && (j.Kw == 0 || (j.lo && j.Nw % 32 == 0 && j.Kw % 16 == 0)),
                   "cn_split_group_launch: job %d: the weight seen by the product must be [32 a, 16 b], got [%d (padded %d), %d]", i, j.Nreal, j.Nw, j.Kw);
        tab.job[i] = SplitGroupJob{j.w, (__bf16 *)j.hi, (__bf16 *)j.lo, j.Nw, j.Kw, j.transpose, j.Nreal, blocks};
        const size_t elems = j.Kw == 0 ? (size_t)j.Nw : (size_t)j.Nw * j.Kw;
        blocks += (int)((elems + 255) / 256);
    }
    hipLaunchKernelGGL(split_bf16_group_kernel, dim3(blocks), dim3(256), 0, st, tab);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_split_bf16(const float *w, int rows, int cols, int transpose, void *hi, void *lo, void *stream)
{
    return cn_split_bf16_padded(w, rows, cols, transpose, 0, hi, lo, stream);
}

extern "C" int cn_linear_fwd(int M, int N, int K, const float *X, int ldx, const float *relu_gate, const void *Whi, const void *Wlo, const float *bias,
                             int act, float *Y, int ldy, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(X && Whi && Wlo && Y && M >= 0, "cn_linear_fwd: bad argument");
    CN_REQUIRE(act == 0 || act == 1, "cn_linear_fwd: act must be 0 (none) or 1 (relu)");
    CN_REQUIRE(ldx >= K && ldy >= N, "cn_linear_fwd: leading dimension smaller than the row length");
    hipStream_t st = (hipStream_t)stream;
    const __bf16 *wh = (const __bf16 *)Whi, *wl = (const __bf16 *)Wlo;
#ifdef CN_G3_KNOCKOUT
    static const int ko = getenv("CN_G3KO") ? atoi(getenv("CN_G3KO")) : 0;
    if (ko && act == 0 && !relu_gate) {
#define KOV(k) if (ko == k) return launch_gemm3p<ACT_NONE, k>(M, N, K, X, ldx, wh, wl, bias, Y, ldy, st, nullptr);
        KOV(1) KOV(4) KOV(5) KOV(8) KOV(9) KOV(12)
#undef KOV
    }
#endif
    return act == 1 ? launch_gemm3p<ACT_RELU>(M, N, K, X, ldx, wh, wl, bias, Y, ldy, st, relu_gate)
                    : launch_gemm3p<ACT_NONE>(M, N, K, X, ldx, wh, wl, bias, Y, ldy, st, relu_gate);
}

extern "C" int cn_linear_fwd_act(int M, int N, int K, const float *X, int ldx, const void *Whi, const void *Wlo, const float *bias, int act,
                                 const float *aux, int ldaux, int relu_from, float *Y, int ldy, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(X && Whi && Wlo && Y && M >= 0, "cn_linear_fwd_act: bad argument");
    CN_REQUIRE(act >= 0 && act <= 4, "cn_linear_fwd_act: act must be 0 (none), 1 (relu), 2 (tanh), 3 (x relu'(aux)) or 4 (x tanh'(aux))");
    CN_REQUIRE(act < 3 || (aux && ldaux >= N), "cn_linear_fwd_act: act %d needs the forward activation aux [M, ldaux >= N]", act);
    CN_REQUIRE(ldx >= K && ldy >= N, "cn_linear_fwd_act: leading dimension smaller than the row length");
    hipStream_t st = (hipStream_t)stream;
    const __bf16 *wh = (const __bf16 *)Whi, *wl = (const __bf16 *)Wlo;
    switch (act) {
    case 0: return launch_gemm3p<ACT_NONE>(M, N, K, X, ldx, wh, wl, bias, Y, ldy, st, nullptr, nullptr, 0, relu_from);
    case 1: return launch_gemm3p<ACT_RELU>(M, N, K, X, ldx, wh, wl, bias, Y, ldy, st, nullptr, nullptr, 0, relu_from);
    case 2: return launch_gemm3p<ACT_TANH>(M, N, K, X, ldx, wh, wl, bias, Y, ldy, st, nullptr, nullptr, 0, relu_from);
    case 3: return launch_gemm3p<ACT_MUL_DRELU>(M, N, K, X, ldx, wh, wl, bias, Y, ldy, st, nullptr, aux, ldaux, relu_from);
    default: return launch_gemm3p<ACT_MUL_DTANH>(M, N, K, X, ldx, wh, wl, bias, Y, ldy, st, nullptr, aux, ldaux, relu_from);
    }
}

// shapes the pipelined TN kernel takes: whole 128-column tiles of dY, enough rows to fill the machine
static bool wgrad_pipelined(int M, int N, int K) { return N % 128 == 0 && K % 128 == 0 && M >= 32768; }
// X columns per workgroup tile of gemm3p_tn_kernel, in units of 128: the widest that divides K.  A 128 x 512 tile reads dY once per
// split instead of twice (the q.k.v gradient: 13.2 -> 11.0 GB of L2 requests per launch) at 484 registers, still one wavefront per SIMD:
// 2.17 -> 1.79 ms stand-alone at 358 k rows (128 x 256 with the same addressing: 2.02).
// ... as long as the 64 splits the launcher allows still give every CU a workgroup (N = 256, K = 512: two 128 x 512 tiles x 64 = half
// the machine, 0.59 ms; four 128 x 256 tiles, 0.48 ms).
static int tn_nb(int N, int K)
{
    int nb = K % 512 == 0 ? 4 : (K % 256 == 0 ? 2 : 1);
    while (nb > 1 && (long long)(N / 128) * (K / (128 * nb)) * 64 < 256) nb >>= 1;
    return nb;
}

extern "C" int cn_linear_wgrad_splits(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0 || N % 64 || K % 128) return 0;
    if (wgrad_pipelined(M, N, K)) { // gemm3p_tn_kernel: one workgroup per CU, tiles of 128 x 512 / 256 / 128 (tn_nb)
        // every XCD (32 CUs, one workgroup each) takes whole splits: s = 8 a with a * tiles a multiple of 32 fills the XCDs in
        // whole rounds; prefer the smallest such s with at least two rounds of work, bounded by 64 partials
        const long long tiles = (long long)(N / 128) * (K / (128 * tn_nb(N, K)));
        long long s = 64;
        for (long long a = 1; a <= 8; ++a)
            if ((a * tiles) % 32 == 0 && a * tiles >= 64) { s = 8 * a; break; }
        const long long chunks = (long long)M / 32;
        if (s > chunks) s = chunks;
        return (int)(s < 1 ? 1 : s) + (M % 32 ? 1 : 0); // + one partial for the last M % 32 rows (gemm3_tn_kernel takes those)
    }
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
    constexpr size_t lds = (size_t)(2 * BM + 2 * 128) * L3_STRIDE * sizeof(__bf16);
    static CnLdsOptIn opt_in; // per device
    int opt_dev;
    if (opt_in.needed(&opt_dev)) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm3_tn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm3_tn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        opt_in.done(opt_dev);
    }
    auto launch_tn = [&](int m, const float *dy, const float *gate, const float *x, int rows_per, int nsplit, float *part, float *dbp) {
        if (gate) hipLaunchKernelGGL(gemm3_tn_kernel<true>, dim3((N + 127) / 128, K / 128, nsplit), dim3(256), lds, st, m, N, K, dy, ldy, gate, x, ldx, rows_per, part, dbp);
        else hipLaunchKernelGGL(gemm3_tn_kernel<false>, dim3((N + 127) / 128, K / 128, nsplit), dim3(256), lds, st, m, N, K, dy, ldy, gate, x, ldx, rows_per, part, dbp);
    };
    int used;
    if (wgrad_pipelined(M, N, K)) {
        // whole tiles of 32 rows on the pipelined kernel, the last M % 32 rows as one more split on the two-barrier kernel
        const int M32 = M / 32 * 32, tail = M - M32, main_splits = splits - (tail ? 1 : 0);
        CN_REQUIRE(main_splits >= 1, "cn_linear_wgrad: splits=%d leaves no room for the row tail (use cn_linear_wgrad_splits)", splits);
        int rows = (M32 + main_splits - 1) / main_splits;
        rows = (rows + 31) / 32 * 32;
        used = (M32 + rows - 1) / rows;
        // the kernel addresses a split's rows through buffer loads: 32-bit byte offsets from the split's first row, 2 GB of records --
        // a load past that returns zero without a fault
        CN_REQUIRE((long long)rows * (ldy > ldx ? ldy : ldx) * 4 < (1LL << 31),
                   "cn_linear_wgrad: %d rows per split x leading dimension %d exceed the 2 GB a split may span (more splits needed)", rows, ldy > ldx ? ldy : ldx);
        constexpr size_t lds2 = (size_t)2 * 2 * 128 * 40 * sizeof(__bf16);
        const int nbk = tn_nb(N, K), tiles = (N / 128) * (K / (128 * nbk));
        const dim3 grid(8 * ((used + 7) / 8) * tiles); // XCD-aware 1-D grid, see the kernel
        if (nbk == 4) {
            if (relu_gate) hipLaunchKernelGGL((gemm3p_tn_kernel<4, true>), grid, dim3(256), lds2, st, M32, N, K, dY, ldy, relu_gate, X, ldx, rows, used, partials, db_partials);
            else hipLaunchKernelGGL((gemm3p_tn_kernel<4, false>), grid, dim3(256), lds2, st, M32, N, K, dY, ldy, relu_gate, X, ldx, rows, used, partials, db_partials);
        } else if (nbk == 2) {
            if (relu_gate) hipLaunchKernelGGL((gemm3p_tn_kernel<2, true>), grid, dim3(256), lds2, st, M32, N, K, dY, ldy, relu_gate, X, ldx, rows, used, partials, db_partials);
            else hipLaunchKernelGGL((gemm3p_tn_kernel<2, false>), grid, dim3(256), lds2, st, M32, N, K, dY, ldy, relu_gate, X, ldx, rows, used, partials, db_partials);
        } else {
            if (relu_gate) hipLaunchKernelGGL((gemm3p_tn_kernel<1, true>), grid, dim3(256), lds2, st, M32, N, K, dY, ldy, relu_gate, X, ldx, rows, used, partials, db_partials);
            else hipLaunchKernelGGL((gemm3p_tn_kernel<1, false>), grid, dim3(256), lds2, st, M32, N, K, dY, ldy, relu_gate, X, ldx, rows, used, partials, db_partials);
        }
        CN_CHECK_LAUNCH();
        if (tail) {
            launch_tn(tail, dY + (size_t)M32 * ldy, relu_gate ? relu_gate + (size_t)M32 * ldy : nullptr, X + (size_t)M32 * ldx, 32, 1,
                      partials + (size_t)used * N * K, db_partials ? db_partials + (size_t)used * N : nullptr);
            CN_CHECK_LAUNCH();
            ++used;
        }
    } else {
        int rows = (M + splits - 1) / splits;
        rows = (rows + BK3 - 1) / BK3 * BK3;
        used = (M + rows - 1) / rows; // <= splits, every split non-empty
        launch_tn(M, dY, relu_gate, X, rows, used, partials, db_partials);
        CN_CHECK_LAUNCH();
    }
    const size_t nk = (size_t)N * K;
    if (db) launch_reduce_partials_pair(nk, (size_t)N, used, partials, dW, db_partials, db, st); // one launch for both sums (same summation orders)
    else launch_reduce_partials(nk, used, partials, dW, st);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Small weight-by-weight products of the update (the folds of policy.py: (q|k|v)_linear o in_proj, out_proj o spatial_linear,
// Ws^T Wt, (actor.0 ; critic.0) o output_linear, their bias images, and the backward of each): C[M,N] = A . B in exact fp32 with
// GENERAL strides for both operands, so that every transposed form of the chain rule is the same launch (dA = dC B^T, dB = A^T dC)
// and a matrix-vector product is the N = 1 case.  At most 512 x 512 x 512 per call, ~40 calls per optimiser step: a 64 x 64 tile per
// workgroup on the exact-fp32 matrix instruction, K tiles of 32 through LDS.  Fixed summation order: deterministic.
// (Round 4 left these on the library's GEMM / GEMV kernels -- the only library products on the training path.)
// ------------------------------------------------------------------------------------------------------------------
namespace {
// 64 x 64 output tile per workgroup, four wavefronts 2 x 2, each a 32 x 32 block on v_mfma_f32_32x32x2_f32 (exact fp32, the k order of
// gemm.h's kernel); both operands go through LDS as [row][k] whatever their strides.
__global__ __launch_bounds__(256) void small_mm_kernel(int M, int N, int K, const float *__restrict__ A, long long sam, long long sak,
                                                       const float *__restrict__ B, long long sbk, long long sbn, float *__restrict__ C)
{
    constexpr int T = 64, KT = 32, LS = 36; // tile, K tile, LDS row stride (144 B: conflict-free 16-byte fragment reads)
    __shared__ __attribute__((aligned(16))) float As[T * LS], Bs[T * LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    // this thread's eight elements of each 64 x 32 operand tile; consecutive lanes walk the operand's unit-stride dimension
    int ar[8], ak[8], bn[8], bk[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int i = tid + 256 * p;                       // 0 .. 2047
        if (sak == 1) { ar[p] = i >> 5; ak[p] = i & 31; } else { ar[p] = i & 63; ak[p] = i >> 6; }   // A tile: row m (64), column k (32)
        if (sbk == 1) { bn[p] = i >> 5; bk[p] = i & 31; } else { bn[p] = i & 63; bk[p] = i >> 6; }   // B tile as [n (64)][k (32)]
    }
    float ra[8], rb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int m = m0 + ar[p], ka = k0 + ak[p];
            ra[p] = (m < M && ka < K) ? A[(long long)m * sam + (long long)ka * sak] : 0.0f;
            const int n = n0 + bn[p], kb = k0 + bk[p];
            rb[p] = (n < N && kb < K) ? B[(long long)kb * sbk + (long long)n * sbn] : 0.0f;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KT) {
        __syncthreads(); // the previous tile is consumed
#pragma unroll
        for (int p = 0; p < 8; ++p) { As[ar[p] * LS + ak[p]] = ra[p]; Bs[bn[p] * LS + bk[p]] = rb[p]; }
        __syncthreads();
        if (k0 + KT < K) fetch(k0 + KT); // the next tile travels while this one is multiplied
#pragma unroll
        for (int g = 0; g < KT / 8; ++g) {
            const f32x4 af = *reinterpret_cast<const f32x4 *>(&As[(wm * 32 + l31) * LS + g * 8 + half * 4]);
            const f32x4 bf = *reinterpret_cast<const f32x4 *>(&Bs[(wn * 32 + l31) * LS + g * 8 + half * 4]);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s4], bf[s4], acc, 0, 0, 0);
        }
    }
    // C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = n0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[r];
    }
}
} // namespace

extern "C" int cn_small_mm(int M, int N, int K, const float *A, int64_t a_stride_m, int64_t a_stride_k, const float *B, int64_t b_stride_k, int64_t b_stride_n,
                           float *C, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(M >= 1 && N >= 1 && K >= 1 && A && B && C, "cn_small_mm: bad argument");
    CN_REQUIRE(M <= 4096 && N <= 4096 && K <= 4096, "cn_small_mm: M=%d N=%d K=%d -- meant for weight-sized operands (<= 4096 per dimension)", M, N, K);
    hipLaunchKernelGGL(small_mm_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, M, N, K, A, (long long)a_stride_m,
                       (long long)a_stride_k, B, (long long)b_stride_k, (long long)b_stride_n, C);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
