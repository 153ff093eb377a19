// gst_train.hip -- one optimiser step's forward + loss + backward of the GST trajectory predictor on gfx950 (cn_gst_train_step).
//
// Reference: gst_updated/scripts/experiments/train.py:107-146 (the loop body: forward, negative log-likelihood, backward) over
// gst_updated/src/gumbel_social_transformer/st_model.py:271-455 (training-time forward: 'faster_lstm', recursive decoding on the mean,
// only_observe_full_period = False) and :62-112 (negative_log_likelihood_full_partial), for the shipped hyper-parameters (embedding 64,
// 8 heads, one NodeEncoderLayer without ghost / edge heads, LSTM 64, 5 observed + 5 predicted steps).  The host mirror of the same
// math in torch ops is crowdnav_prediction_attngraph_amd/gst_train.py (forward_train / negative_log_likelihood_full_partial).
//
// Mapping.  A sequence is <= 64 pedestrians x 10 steps through a 67 k-parameter model: one WORKGROUP per sequence (B sequences of a
// batch per launch), every activation of the nine encoder-layer passes and nine LSTM steps kept in a per-sequence scratch slab for
// the hand-derived reverse pass, every parameter gradient accumulated in a per-workgroup slab (no atomics, fixed summation orders)
// and reduced over the batch by a second launch.  Plain fp32 FMA arithmetic: the products are 64 x 64 x 192 at most.
// Dropout (the reference's four sites: attention probabilities, out_proj output, FFN hidden, FFN output; p = 0.1 in training) draws its
// masks from a counter-based hash of (seed, layer pass, site, element), regenerated in the backward pass -- its own stream, not torch's.
#include "common.h"

#include <cmath>

namespace {

constexpr int GT = 5, GP = 5, TT = GT + GP;   // observed / predicted steps
constexpr int NCALL = GT + GP - 1;            // encoder-layer passes: 5 observed slices + 4 decode steps
constexpr int NSTEP = GT + GP - 1;            // LSTM steps
constexpr int NT = 256;                       // threads per workgroup
constexpr int NPARAM = 20;
constexpr int PSIZE[NPARAM] = {128, 64, 192 * 64, 192, 64 * 64, 64, 64, 64, 64, 64, 128 * 64, 128, 64 * 128, 64, 256 * 64, 256 * 64, 256, 256, 320, 5};
enum { P_EW = 0, P_EB, P_INW, P_INB, P_OW, P_OB, P_NW, P_NB, P_N1W, P_N1B, P_L1W, P_L1B, P_L2W, P_L2B, P_WIH, P_WHH, P_BIH, P_BHH, P_HW, P_HB };
constexpr int param_total()
{
    int s = 0;
    for (int i = 0; i < NPARAM; ++i) s += PSIZE[i];
    return s;
}
constexpr int NPARAMS = param_total(); // 67 269

struct Wts { const float *p[NPARAM]; };
struct Grd { float *p[NPARAM]; };

// ---- per-sequence scratch layout (floats), N = pedestrians (padded count of the batch) ----
struct Lay {
    int N;
    // one encoder-layer pass
    int c_x2, c_mean0, c_rstd0, c_xh0, c_n0, c_qkv, c_p, c_s, c_o, c_x1, c_mean1, c_rstd1, c_xh1, c_n1, c_f, c_out, call_size;
    // one LSTM step
    int s_x, s_hp, s_cp, s_g, s_tc, step_size;
    // whole sequence
    int calls, steps, hdec, raw, work, total;
    // work buffers inside `work`
    int w_dx, w_dh, w_dc, w_dqkv, w_ds, w_do, w_t64a, w_t64b, w_t128, w_dg, w_dx2, w_hcur, w_ccur, w_xs, w_xsample, w_dmu, work_size;
};
__host__ __device__ inline Lay make_lay(int N)
{
    Lay L{};
    L.N = N;
    int o = 0;
    auto f = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    L.c_x2 = f(N * 2); L.c_mean0 = f(N); L.c_rstd0 = f(N); L.c_xh0 = f(N * 64); L.c_n0 = f(N * 64); L.c_qkv = f(N * 192); L.c_p = f(8 * N * N); L.c_s = f(8 * N);
    L.c_o = f(N * 64); L.c_x1 = f(N * 64); L.c_mean1 = f(N); L.c_rstd1 = f(N); L.c_xh1 = f(N * 64); L.c_n1 = f(N * 64); L.c_f = f(N * 128); L.c_out = f(N * 64);
    L.call_size = o;
    o = 0;
    L.s_x = f(N * 64); L.s_hp = f(N * 64); L.s_cp = f(N * 64); L.s_g = f(N * 256); L.s_tc = f(N * 64);
    L.step_size = o;
    o = 0;
    L.w_dx = f(N * 64); L.w_dh = f(N * 64); L.w_dc = f(N * 64); L.w_dqkv = f(N * 192); L.w_ds = f(8 * N * N); L.w_do = f(N * 64); L.w_t64a = f(N * 64); L.w_t64b = f(N * 64);
    L.w_t128 = f(N * 128); L.w_dg = f(N * 256); L.w_dx2 = f(N * 2); L.w_hcur = f(N * 64); L.w_ccur = f(N * 64); L.w_xs = f(N * 64); L.w_xsample = f(N * 2); L.w_dmu = f(N * 2);
    L.work_size = o;
    o = 0;
    L.calls = f(NCALL * L.call_size); L.steps = f(NSTEP * L.step_size); L.hdec = f(GP * N * 64); L.raw = f(GP * N * 5); L.work = f(L.work_size);
    L.total = o;
    return L;
}

// ---- dropout: counter-based hash -> keep / scale (torch's F.dropout semantics: zero with probability p, survivors x 1 / (1 - p)) ----
__device__ __forceinline__ float drop_scale(unsigned long long seed, int call, int site, unsigned idx, float p)
{
    if (p <= 0.0f) return 1.0f;
    unsigned long long x = seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(call * 4 + site + 1)) ^ ((unsigned long long)idx * 0xD1B54A32D192ED03ull);
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    const float u = (float)(x >> 40) * (1.0f / 16777216.0f);
    return u < p ? 0.0f : 1.0f / (1.0f - p);
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// y[r][f] = b[f] + sum_k W[f][k] x[r][k]
__device__ void lin_fwd(int N, int K, int F, const float *x, int ldx, const float *__restrict__ W, const float *__restrict__ b, float *y, int ldy)
{
    for (int idx = threadIdx.x; idx < N * F; idx += NT) {
        const int r = idx / F, f = idx - r * F;
        const float *w = W + (size_t)f * K, *xr = x + (size_t)r * ldx;
        float acc = b ? b[f] : 0.0f;
        for (int k = 0; k < K; ++k) acc += w[k] * xr[k];
        y[(size_t)r * ldy + f] = acc;
    }
}
// dx[r][k] (+)= sum_f dy[r][f] W[f][k]
__device__ void lin_bwd_x(int N, int K, int F, const float *dy, int ldy, const float *__restrict__ W, float *dx, int ldx, bool accumulate)
{
    for (int idx = threadIdx.x; idx < N * K; idx += NT) {
        const int r = idx / K, k = idx - r * K;
        const float *d = dy + (size_t)r * ldy;
        float acc = 0.0f;
        for (int f = 0; f < F; ++f) acc += d[f] * W[(size_t)f * K + k];
        float *o = dx + (size_t)r * ldx + k;
        *o = accumulate ? *o + acc : acc;
    }
}
// gW[f][k] += sum_r dy[r][f] x[r][k];  gb[f] += sum_r dy[r][f]   (this workgroup's own slab: plain read-modify-write)
__device__ void lin_bwd_w(int N, int K, int F, const float *dy, int ldy, const float *x, int ldx, float *gW, float *gb)
{
    for (int idx = threadIdx.x; idx < F * K; idx += NT) {
        const int f = idx / K, k = idx - f * K;
        float acc = 0.0f;
        for (int r = 0; r < N; ++r) acc += dy[(size_t)r * ldy + f] * x[(size_t)r * ldx + k];
        gW[idx] += acc;
    }
    if (gb)
        for (int f = threadIdx.x; f < F; f += NT) {
            float acc = 0.0f;
            for (int r = 0; r < N; ++r) acc += dy[(size_t)r * ldy + f];
            gb[f] += acc;
        }
}
// LayerNorm over 64 features, one wavefront per row (lane = feature): y = xhat g + b, xhat / mean / rstd kept for the backward
__device__ void ln_fwd(int N, const float *x, const float *__restrict__ g, const float *__restrict__ b, float *y, float *xhat, float *mean, float *rstd,
                       const float *rowmask)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < N; r += NT / 64) {
        const float v = x[r * 64 + lane];
        const float m = wv_sum(v) * (1.0f / 64.0f);
        const float d = v - m;
        const float var = wv_sum(d * d) * (1.0f / 64.0f);
        const float rs = 1.0f / sqrtf(var + 1e-5f);
        const float xh = d * rs;
        xhat[r * 64 + lane] = xh;
        y[r * 64 + lane] = (xh * g[lane] + b[lane]) * (rowmask ? rowmask[r] : 1.0f);
        if (lane == 0) { mean[r] = m; rstd[r] = rs; }
    }
}
// dx = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)), dxh = dy g (dy already times the row mask); gg += sum_r dy xhat, gb += sum_r dy
__device__ void ln_bwd(int N, const float *dy, const float *xhat, const float *rstd, const float *__restrict__ g, float *dx, bool accumulate, float *gg, float *gb,
                       float *lds /* >= 2 * 4 * 64 floats */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float sg = 0.0f, sb = 0.0f;
    for (int r = wave; r < N; r += NT / 64) {
        const float d = dy[r * 64 + lane], xh = xhat[r * 64 + lane];
        sg += d * xh; sb += d;
        const float dxh = d * g[lane];
        const float m1 = wv_sum(dxh) * (1.0f / 64.0f), m2 = wv_sum(dxh * xh) * (1.0f / 64.0f);
        const float v = rstd[r] * (dxh - m1 - xh * m2);
        float *o = dx + r * 64 + lane;
        *o = accumulate ? *o + v : v;
    }
    lds[wave * 64 + lane] = sg; lds[256 + wave * 64 + lane] = sb;
    __syncthreads();
    if (wave == 0) {
        gg[lane] += (lds[lane] + lds[64 + lane]) + (lds[128 + lane] + lds[192 + lane]);
        gb[lane] += (lds[256 + lane] + lds[320 + lane]) + (lds[384 + lane] + lds[448 + lane]);
    }
    __syncthreads();
}

// ---- NodeEncoderLayer forward on N rows with the 0/1 presence vector m (attention mask m_i m_j): x2 [N,2] -> out [N,64] ----
__device__ void layer_fwd(const Lay &L, float *C, const Wts &W, const float *x2, const float *m, int call, float p_drop, unsigned long long seed)
{
    const int N = L.N;
    float *X2 = C + L.c_x2, *XH0 = C + L.c_xh0, *N0 = C + L.c_n0, *QKV = C + L.c_qkv, *Pp = C + L.c_p, *S = C + L.c_s, *O = C + L.c_o, *X1 = C + L.c_x1,
          *XH1 = C + L.c_xh1, *N1 = C + L.c_n1, *Ff = C + L.c_f, *OUT = C + L.c_out;
    for (int i = threadIdx.x; i < N * 2; i += NT) X2[i] = x2[i];
    __syncthreads();
    // node_embedding (2 -> 64) into OUT (scratch), LayerNorm(norm_node) * ped -> N0
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) {
        const int r = idx >> 6, f = idx & 63;
        OUT[idx] = W.p[P_EW][2 * f] * X2[2 * r] + W.p[P_EW][2 * f + 1] * X2[2 * r + 1] + W.p[P_EB][f];
    }
    __syncthreads();
    ln_fwd(N, OUT, W.p[P_NW], W.p[P_NB], N0, XH0, C + L.c_mean0, C + L.c_rstd0, m);
    __syncthreads();
    lin_fwd(N, 64, 192, N0, 64, W.p[P_INW], W.p[P_INB], QKV, 192);
    __syncthreads();
    // attention per (row i, head h): softmax over all j, times the mask, renormalised (mha.py:236-242), dropout, times v
    for (int ih = threadIdx.x; ih < N * 8; ih += NT) {
        const int i = ih >> 3, h = ih & 7;
        const float *q = QKV + i * 192 + h * 8;
        float *p = Pp + ((size_t)h * N + i) * N;
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) {
            const float *k = QKV + j * 192 + 64 + h * 8;
            float s = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) s += (q[d] * 0.35355339059327373f) * k[d];
            p[j] = s;
            mx = fmaxf(mx, s);
        }
        float Z = 0.0f;
        for (int j = 0; j < N; ++j) { const float e = expf(p[j] - mx); p[j] = e; Z += e; }
        float Sm = 0.0f;
        for (int j = 0; j < N; ++j) { const float pj = p[j] / Z; p[j] = pj; Sm += pj * (m[i] * m[j]); }
        S[h * N + i] = Sm;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float inv = 1.0f / (Sm + 1e-10f);
        for (int j = 0; j < N; ++j) {
            const float pr = p[j] * (m[i] * m[j]) * inv * drop_scale(seed, call, 0, (unsigned)((h * N + i) * N + j), p_drop);
            const float *v = QKV + j * 192 + 128 + h * 8;
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] += pr * v[d];
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) O[i * 64 + h * 8 + d] = o[d];
    }
    __syncthreads();
    // x1 = n0 + dropout(out_proj(o))
    lin_fwd(N, 64, 64, O, 64, W.p[P_OW], W.p[P_OB], X1, 64);
    __syncthreads();
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) X1[idx] = N0[idx] + X1[idx] * drop_scale(seed, call, 1, (unsigned)idx, p_drop);
    __syncthreads();
    ln_fwd(N, X1, W.p[P_N1W], W.p[P_N1B], N1, XH1, C + L.c_mean1, C + L.c_rstd1, nullptr);
    __syncthreads();
    lin_fwd(N, 64, 128, N1, 64, W.p[P_L1W], W.p[P_L1B], Ff, 128);
    __syncthreads();
    for (int idx = threadIdx.x; idx < N * 128; idx += NT) Ff[idx] = fmaxf(Ff[idx], 0.0f) * drop_scale(seed, call, 2, (unsigned)idx, p_drop);
    __syncthreads();
    lin_fwd(N, 128, 64, Ff, 128, W.p[P_L2W], W.p[P_L2B], OUT, 64);
    __syncthreads();
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) OUT[idx] = X1[idx] + OUT[idx] * drop_scale(seed, call, 3, (unsigned)idx, p_drop);
    __syncthreads();
}

// ---- its backward: d_out [N,64] (gradient of the layer output) -> parameter gradients (+=) and d_x2 [N,2] ----
__device__ void layer_bwd(const Lay &L, float *C, const Wts &W, const Grd &G, float *Wk, const float *d_out, const float *m, int call, float p_drop,
                          unsigned long long seed, float *d_x2, float *lds)
{
    const int N = L.N;
    const float *X2 = C + L.c_x2, *XH0 = C + L.c_xh0, *N0 = C + L.c_n0, *QKV = C + L.c_qkv, *Pp = C + L.c_p, *S = C + L.c_s, *O = C + L.c_o,
                *XH1 = C + L.c_xh1, *N1 = C + L.c_n1, *Ff = C + L.c_f;
    float *T64A = Wk + L.w_t64a, *T64B = Wk + L.w_t64b, *T128 = Wk + L.w_t128, *DQKV = Wk + L.w_dqkv, *DS = Wk + L.w_ds, *DO = Wk + L.w_do;
    // out = x1 + drop3(W2 f + b2):  T64A = d(W2 f + b2) = d_out * mask3 ;  d_x1 starts as d_out (T64B)
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) {
        T64A[idx] = d_out[idx] * drop_scale(seed, call, 3, (unsigned)idx, p_drop);
        T64B[idx] = d_out[idx];
    }
    __syncthreads();
    lin_bwd_w(N, 128, 64, T64A, 64, Ff, 128, G.p[P_L2W], G.p[P_L2B]);
    lin_bwd_x(N, 128, 64, T64A, 64, W.p[P_L2W], T128, 128, false);       // d f (post-dropout)
    __syncthreads();
    // f = relu(pre) * mask2: d pre = d f * mask2 * [pre > 0]; f > 0 <=> pre > 0 and kept
    for (int idx = threadIdx.x; idx < N * 128; idx += NT) T128[idx] = Ff[idx] > 0.0f ? T128[idx] * drop_scale(seed, call, 2, (unsigned)idx, p_drop) : 0.0f;
    __syncthreads();
    lin_bwd_w(N, 64, 128, T128, 128, N1, 64, G.p[P_L1W], G.p[P_L1B]);
    lin_bwd_x(N, 64, 128, T128, 128, W.p[P_L1W], T64A, 64, false);       // d n1
    __syncthreads();
    ln_bwd(N, T64A, XH1, C + L.c_rstd1, W.p[P_N1W], T64B, true, G.p[P_N1W], G.p[P_N1B], lds);   // d x1 += LN1 backward
    // x1 = n0 + drop1(Wo o + bo): T64A = d(Wo o + bo) = d_x1 * mask1
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) T64A[idx] = T64B[idx] * drop_scale(seed, call, 1, (unsigned)idx, p_drop);
    __syncthreads();
    lin_bwd_w(N, 64, 64, T64A, 64, O, 64, G.p[P_OW], G.p[P_OB]);
    lin_bwd_x(N, 64, 64, T64A, 64, W.p[P_OW], DO, 64, false);            // d o
    __syncthreads();
    // attention backward, pass A per (i, h): d s_ij -> DS, d q_i
    for (int ih = threadIdx.x; ih < N * 8; ih += NT) {
        const int i = ih >> 3, h = ih & 7;
        const float *p = Pp + ((size_t)h * N + i) * N;
        float *ds = DS + ((size_t)h * N + i) * N;
        const float *dofs = DO + i * 64 + h * 8;
        const float inv = 1.0f / (S[h * N + i] + 1e-10f);
        // d pr_j (through the dropout mask) and A = sum_j d pr_j pr_j
        float A = 0.0f;
        for (int j = 0; j < N; ++j) {
            const float *v = QKV + j * 192 + 128 + h * 8;
            float d = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += dofs[e] * v[e];
            d *= drop_scale(seed, call, 0, (unsigned)((h * N + i) * N + j), p_drop);
            ds[j] = d;
            A += d * (p[j] * (m[i] * m[j]) * inv);
        }
        // d pm_j = (d pr_j - A) / (S + eps); d p_j = d pm_j * mask_ij; softmax backward
        float Bs = 0.0f;
        for (int j = 0; j < N; ++j) {
            const float dp = (ds[j] - A) * inv * (m[i] * m[j]);
            ds[j] = dp;
            Bs += p[j] * dp;
        }
        float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < N; ++j) {
            const float dsj = p[j] * (ds[j] - Bs);
            ds[j] = dsj;
            const float *k = QKV + j * 192 + 64 + h * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) dq[e] += dsj * k[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) DQKV[i * 192 + h * 8 + e] = dq[e] * 0.35355339059327373f;
    }
    __syncthreads();
    // pass B per (j, h): d k_j = scale sum_i d s_ij q_i ;  d v_j = sum_i pr_ij(dropped) d o_i
    for (int jh = threadIdx.x; jh < N * 8; jh += NT) {
        const int j = jh >> 3, h = jh & 7;
        float dk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < N; ++i) {
            const float dsij = DS[((size_t)h * N + i) * N + j];
            const float pr = Pp[((size_t)h * N + i) * N + j] * (m[i] * m[j]) / (S[h * N + i] + 1e-10f) * drop_scale(seed, call, 0, (unsigned)((h * N + i) * N + j), p_drop);
            const float *q = QKV + i * 192 + h * 8, *dofs = DO + i * 64 + h * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { dk[e] += dsij * q[e]; dv[e] += pr * dofs[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { DQKV[j * 192 + 64 + h * 8 + e] = dk[e] * 0.35355339059327373f; DQKV[j * 192 + 128 + h * 8 + e] = dv[e]; }
    }
    __syncthreads();
    lin_bwd_w(N, 64, 192, DQKV, 192, N0, 64, G.p[P_INW], G.p[P_INB]);
    lin_bwd_x(N, 64, 192, DQKV, 192, W.p[P_INW], T64B, 64, true);        // d n0 = d x1 (residual) + in_proj backward
    __syncthreads();
    // n0 = LN0(e) * ped: d(LN0 output) = d n0 * ped
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) T64B[idx] *= m[idx >> 6];
    __syncthreads();
    ln_bwd(N, T64B, XH0, C + L.c_rstd0, W.p[P_NW], T64A, false, G.p[P_NW], G.p[P_NB], lds);   // d e
    // e = We x2 + be
    for (int idx = threadIdx.x; idx < 128; idx += NT) {
        const int f = idx >> 1, c = idx & 1;
        float acc = 0.0f;
        for (int r = 0; r < N; ++r) acc += T64A[r * 64 + f] * X2[2 * r + c];
        G.p[P_EW][idx] += acc;
    }
    for (int f = threadIdx.x; f < 64; f += NT) {
        float acc = 0.0f;
        for (int r = 0; r < N; ++r) acc += T64A[r * 64 + f];
        G.p[P_EB][f] += acc;
    }
    if (d_x2)
        for (int idx = threadIdx.x; idx < N * 2; idx += NT) {
            const int r = idx >> 1, c = idx & 1;
            float acc = 0.0f;
            for (int f = 0; f < 64; ++f) acc += T64A[r * 64 + f] * W.p[P_EW][2 * f + c];
            d_x2[idx] = acc;
        }
    __syncthreads();
}

// ---- LSTM cell (PyTorch gate order i, f, g, o): x, h, c [N,64] -> h', c'; everything the backward needs into the step's slab ----
__device__ void lstm_fwd(const Lay &L, float *Sx, const Wts &W, const float *x, float *h, float *c)
{
    const int N = L.N;
    float *SX = Sx + L.s_x, *HP = Sx + L.s_hp, *CP = Sx + L.s_cp, *GG = Sx + L.s_g, *TC = Sx + L.s_tc;
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) { SX[idx] = x[idx]; HP[idx] = h[idx]; CP[idx] = c[idx]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < N * 256; idx += NT) {
        const int r = idx >> 8, g = idx & 255;
        const float *wi = W.p[P_WIH] + g * 64, *wh = W.p[P_WHH] + g * 64, *xr = SX + r * 64, *hr = HP + r * 64;
        float acc = W.p[P_BIH][g] + W.p[P_BHH][g];
        for (int k = 0; k < 64; ++k) acc += wi[k] * xr[k];
        for (int k = 0; k < 64; ++k) acc += wh[k] * hr[k];
        GG[idx] = (g >= 128 && g < 192) ? tanhf(acc) : sigm(acc);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) {
        const int r = idx >> 6, d = idx & 63;
        const float *g = GG + r * 256;
        const float cn = g[64 + d] * CP[idx] + g[d] * g[128 + d];
        const float tc = tanhf(cn);
        TC[idx] = tc;
        c[idx] = cn;
        h[idx] = g[192 + d] * tc;
    }
    __syncthreads();
}
// dh, dc: gradients of (h', c') on entry, of (h, c) on exit; dx [N,64] out; weight gradients +=
__device__ void lstm_bwd(const Lay &L, const float *Sx, const Wts &W, const Grd &G, float *Wk, float *dh, float *dc, float *dx)
{
    const int N = L.N;
    const float *SX = Sx + L.s_x, *HP = Sx + L.s_hp, *CP = Sx + L.s_cp, *GG = Sx + L.s_g, *TC = Sx + L.s_tc;
    float *DG = Wk + L.w_dg;
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) {
        const int r = idx >> 6, d = idx & 63;
        const float *g = GG + r * 256;
        const float i = g[d], f = g[64 + d], gg = g[128 + d], o = g[192 + d], tc = TC[idx];
        const float dhn = dh[idx];
        const float dcn = dc[idx] + dhn * o * (1.0f - tc * tc);
        float *dg = DG + r * 256;
        dg[d] = dcn * gg * i * (1.0f - i);
        dg[64 + d] = dcn * CP[idx] * f * (1.0f - f);
        dg[128 + d] = dcn * i * (1.0f - gg * gg);
        dg[192 + d] = dhn * tc * o * (1.0f - o);
        dc[idx] = dcn * f;
    }
    __syncthreads();
    lin_bwd_w(N, 64, 256, DG, 256, SX, 64, G.p[P_WIH], G.p[P_BIH]);
    lin_bwd_w(N, 64, 256, DG, 256, HP, 64, G.p[P_WHH], G.p[P_BHH]);
    lin_bwd_x(N, 64, 256, DG, 256, W.p[P_WIH], dx, 64, false);
    lin_bwd_x(N, 64, 256, DG, 256, W.p[P_WHH], dh, 64, false);
    __syncthreads();
}

// count of valid (predicted step, pedestrian) pairs of the whole batch: the loss's denominator (train.py:133)
__global__ void gst_count_kernel(int B, int N, const float *__restrict__ lm, float *__restrict__ count)
{
    __shared__ float red[NT];
    float s = 0.0f;
    for (int idx = threadIdx.x; idx < B * N * GP; idx += NT) {
        const int b = idx / (N * GP), rem = idx - b * N * GP, n = rem / GP, tt = rem - n * GP;
        const float *l = lm + ((size_t)b * N + n) * TT;
        s += l[GT + tt] * l[GT - 1];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) count[0] = red[0];
}

__global__ __launch_bounds__(NT) void gst_train_kernel(int N, const float *__restrict__ v_obs, const float *__restrict__ v_pred, const float *__restrict__ lm_all, Wts W,
                                                        float p_drop, unsigned long long seed, float *__restrict__ scratch, float *__restrict__ grad_slabs,
                                                        const float *__restrict__ count, float *__restrict__ loss_part, float *__restrict__ gauss_out)
{
    __shared__ float lds[768]; // [0,64) row mask of the current pass | [64,128) lm_fp | [128,132) loss partials | [160,672) LayerNorm-backward sums
    __shared__ float s_lm[64 * TT];
    const int b = blockIdx.x;
    const Lay L = make_lay(N);
    float *base = scratch + (size_t)b * L.total;
    float *Wk = base + L.work;
    float *gs = grad_slabs + (size_t)b * NPARAMS;
    Grd G;
    {
        int o = 0;
        for (int i = 0; i < NPARAM; ++i) { G.p[i] = gs + o; o += PSIZE[i]; }
    }
    for (int i = threadIdx.x; i < NPARAMS; i += NT) gs[i] = 0.0f;
    for (int i = threadIdx.x; i < N * TT; i += NT) s_lm[i] = lm_all[(size_t)b * N * TT + i];
    __syncthreads();
    const unsigned long long sd = seed + 0x632BE59BD9B4E019ull * (unsigned long long)(b + 1);
    float *H = Wk + L.w_hcur, *Cc = Wk + L.w_ccur, *XS = Wk + L.w_xs, *XSMP = Wk + L.w_xsample, *MK = lds + 0; // MK: per-row mask of the current pass (<= 64 floats)
    float *MFP = lds + 64;                                                                                          // lm_fp (<= 64 floats)
    for (int r = threadIdx.x; r < N; r += NT) MFP[r] = s_lm[r * TT + GT - 1];
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) { H[idx] = 0.0f; Cc[idx] = 0.0f; }
    __syncthreads();
    // ================= forward =================
    // observation period: encoder layer on every observed slice, LSTM over time
    for (int t = 0; t < GT; ++t) {
        for (int r = threadIdx.x; r < N; r += NT) MK[r] = s_lm[r * TT + t];
        __syncthreads();
        float *C = base + L.calls + (size_t)t * L.call_size;
        layer_fwd(L, C, W, v_obs + ((size_t)b * GT + t) * N * 2, MK, t, p_drop, sd);
        for (int idx = threadIdx.x; idx < N * 64; idx += NT) XS[idx] = C[L.c_out + idx] * MK[idx >> 6];
        __syncthreads();
        lstm_fwd(L, base + L.steps + (size_t)t * L.step_size, W, XS, H, Cc);
    }
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) { H[idx] *= MFP[idx >> 6]; Cc[idx] *= MFP[idx >> 6]; }
    __syncthreads();
    // prediction period: head on the state, recursive decoding on the mean
    for (int tt = 0; tt < GP; ++tt) {
        if (tt > 0) {
            float *C = base + L.calls + (size_t)(GT + tt - 1) * L.call_size;
            layer_fwd(L, C, W, XSMP, MFP, GT + tt - 1, p_drop, sd);
            for (int idx = threadIdx.x; idx < N * 64; idx += NT) XS[idx] = C[L.c_out + idx] * MFP[idx >> 6];
            __syncthreads();
            float *Sx = base + L.steps + (size_t)(GT + tt - 1) * L.step_size;
            // (the cell overwrites H / Cc with h', c'; the blend needs the old state: it is in the step's slab)
            lstm_fwd(L, Sx, W, XS, H, Cc);
            for (int idx = threadIdx.x; idx < N * 64; idx += NT) {
                const float mk = MFP[idx >> 6];
                H[idx] = H[idx] * mk + Sx[L.s_hp + idx] * (1.0f - mk);
                Cc[idx] = Cc[idx] * mk + Sx[L.s_cp + idx] * (1.0f - mk);
            }
            __syncthreads();
        }
        float *HD = base + L.hdec + (size_t)tt * N * 64, *RAW = base + L.raw + (size_t)tt * N * 5;
        for (int idx = threadIdx.x; idx < N * 64; idx += NT) HD[idx] = H[idx];
        __syncthreads();
        lin_fwd(N, 64, 5, HD, 64, W.p[P_HW], W.p[P_HB], RAW, 5);
        __syncthreads();
        for (int idx = threadIdx.x; idx < N * 2; idx += NT) XSMP[idx] = RAW[(idx >> 1) * 5 + (idx & 1)] * MFP[idx >> 1];
        __syncthreads();
    }
    // ================= loss (st_model.py:62-112) and its gradient w.r.t. the raw head outputs =================
    const float inv_count = 1.0f / fmaxf(count[0], 1e-20f);
    float lsum = 0.0f;
    // d raw [GP][N][5] overwrites RAW's place in a work buffer: reuse w_dg (N * 256 >= GP * N * 5)
    float *DRAW = Wk + L.w_dg;
    for (int idx = threadIdx.x; idx < GP * N; idx += NT) {
        const int tt = idx / N, n = idx - tt * N;
        const float *raw = base + L.raw + ((size_t)tt * N + n) * 5;
        const float M = s_lm[n * TT + GT + tt] * MFP[n];
        const float mux = raw[0], muy = raw[1], sx = expf(raw[2]), sy = expf(raw[3]), rho = tanhf(raw[4]);
        if (gauss_out) {
            float *go = gauss_out + (((size_t)b * GP + tt) * N + n) * 5;
            go[0] = mux; go[1] = muy; go[2] = sx; go[3] = sy; go[4] = rho;
        }
        float *dr = DRAW + idx * 5;
        if (M > 0.0f) {
            const float *xt = v_pred + (((size_t)b * GP + tt) * N + n) * 2;
            const float nx = (xt[0] - mux) / sx, ny = (xt[1] - muy) / sy;
            const float a = 1.0f - rho * rho;
            const float Q = nx * nx - 2.0f * rho * nx * ny + ny * ny;
            lsum += 0.5f * logf(a) + logf(sx) + logf(sy) + Q / (2.0f * a);
            dr[0] = -(nx - rho * ny) / (a * sx) * inv_count;
            dr[1] = -(ny - rho * nx) / (a * sy) * inv_count;
            dr[2] = (1.0f - nx * (nx - rho * ny) / a) * inv_count;
            dr[3] = (1.0f - ny * (ny - rho * nx) / a) * inv_count;
            dr[4] = (-rho - nx * ny + Q * rho / a) * inv_count;
        } else {
            dr[0] = dr[1] = dr[2] = dr[3] = dr[4] = 0.0f;
        }
    }
    {
        lsum = wv_sum(lsum);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) lds[128 + (threadIdx.x >> 6)] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) loss_part[b] = (lds[128] + lds[129]) + (lds[130] + lds[131]);
        __syncthreads();
    }
    // ================= backward =================
    float *DH = Wk + L.w_dh, *DC = Wk + L.w_dc, *DX = Wk + L.w_dx, *DX2 = Wk + L.w_dx2, *DMU = Wk + L.w_dmu;
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) { DH[idx] = 0.0f; DC[idx] = 0.0f; }
    for (int idx = threadIdx.x; idx < N * 2; idx += NT) DMU[idx] = 0.0f; // gradient into x_sample of this step from the NEXT step's encoder pass
    __syncthreads();
    for (int tt = GP - 1; tt >= 0; --tt) {
        // head: raw = Wh h + bh; d raw[0:2] += d x_sample * lm_fp
        float *dr = DRAW + (size_t)tt * N * 5;
        for (int idx = threadIdx.x; idx < N * 2; idx += NT) dr[(idx >> 1) * 5 + (idx & 1)] += DMU[idx] * MFP[idx >> 1];
        __syncthreads();
        const float *HD = base + L.hdec + (size_t)tt * N * 64;
        lin_bwd_w(N, 64, 5, dr, 5, HD, 64, G.p[P_HW], G.p[P_HB]);
        lin_bwd_x(N, 64, 5, dr, 5, W.p[P_HW], DH, 64, true);
        __syncthreads();
        if (tt > 0) {
            // h = h' mk + h_old (1 - mk) (and c alike): split the gradient; the cell's backward turns (d h', d c') into (d h_old, d c_old) contributions
            float *T1 = Wk + L.w_t64a, *T2 = Wk + L.w_t64b; // keep the pass-through parts while lstm_bwd overwrites DH / DC
            for (int idx = threadIdx.x; idx < N * 64; idx += NT) {
                const float mk = MFP[idx >> 6];
                T1[idx] = DH[idx] * (1.0f - mk); T2[idx] = DC[idx] * (1.0f - mk);
                DH[idx] *= mk; DC[idx] *= mk;
            }
            __syncthreads();
            // (lstm_bwd uses w_dg, which holds DRAW: the rows of steps < tt are still needed -> move them out of the way first)
            float *DRAWS = Wk + L.w_hcur; // (the forward's running state is dead by now; layer_bwd below uses w_ds / w_t128 / w_do itself)
            for (int idx = threadIdx.x; idx < tt * N * 5; idx += NT) DRAWS[idx] = DRAW[idx];
            __syncthreads();
            // T1 / T2 live in w_t64a / w_t64b, which layer_bwd uses: park them in w_xs / w_do
            float *P1 = Wk + L.w_xs, *P2 = Wk + L.w_do;
            for (int idx = threadIdx.x; idx < N * 64; idx += NT) { P1[idx] = T1[idx]; P2[idx] = T2[idx]; }
            __syncthreads();
            lstm_bwd(L, base + L.steps + (size_t)(GT + tt - 1) * L.step_size, W, G, Wk, DH, DC, DX);
            for (int idx = threadIdx.x; idx < N * 64; idx += NT) { DH[idx] += P1[idx]; DC[idx] += P2[idx]; DX[idx] *= MFP[idx >> 6]; } // xt = layer_out * mk
            __syncthreads();
            layer_bwd(L, base + L.calls + (size_t)(GT + tt - 1) * L.call_size, W, G, Wk, DX, MFP, GT + tt - 1, p_drop, sd, DX2, lds + 160);
            for (int idx = threadIdx.x; idx < N * 2; idx += NT) DMU[idx] = DX2[idx];
            for (int idx = threadIdx.x; idx < tt * N * 5; idx += NT) DRAW[idx] = DRAWS[idx];
            __syncthreads();
        }
    }
    // h, c *= lm_fp after the observation period
    for (int idx = threadIdx.x; idx < N * 64; idx += NT) { DH[idx] *= MFP[idx >> 6]; DC[idx] *= MFP[idx >> 6]; }
    __syncthreads();
    for (int t = GT - 1; t >= 0; --t) {
        for (int r = threadIdx.x; r < N; r += NT) MK[r] = s_lm[r * TT + t];
        __syncthreads();
        lstm_bwd(L, base + L.steps + (size_t)t * L.step_size, W, G, Wk, DH, DC, DX);
        for (int idx = threadIdx.x; idx < N * 64; idx += NT) DX[idx] *= MK[idx >> 6];
        __syncthreads();
        layer_bwd(L, base + L.calls + (size_t)t * L.call_size, W, G, Wk, DX, MK, t, p_drop, sd, nullptr, lds + 160);
    }
}

// grads[i] = sum over the batch's workgroups of their slabs (fixed order); loss = sum of partial losses / count
__global__ void gst_grad_reduce_kernel(int B, const float *__restrict__ slabs, Grd G, const float *__restrict__ loss_part, const float *__restrict__ count,
                                       float *__restrict__ loss_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NPARAMS) {
        float acc = 0.0f;
        for (int b = 0; b < B; ++b) acc += slabs[(size_t)b * NPARAMS + i];
        int o = i, k = 0;
        while (o >= PSIZE[k]) { o -= PSIZE[k]; ++k; }
        G.p[k][o] = acc;
    }
    if (i == 0) {
        float s = 0.0f;
        for (int b = 0; b < B; ++b) s += loss_part[b];
        loss_out[0] = s / fmaxf(count[0], 1e-20f);
        loss_out[1] = count[0];
    }
}

struct TrainWs { size_t scratch, slabs, count, loss_part, total; };
TrainWs train_ws(int B, int N)
{
    TrainWs w{};
    size_t off = 0;
    auto f = [&](size_t n) { size_t o = off; off += (n * 4 + 255) & ~size_t(255); return o; };
    w.scratch = f((size_t)B * make_lay(N).total); w.slabs = f((size_t)B * NPARAMS); w.count = f(4); w.loss_part = f((size_t)B);
    w.total = off;
    return w;
}

} // namespace

extern "C" int64_t cn_gst_train_workspace_bytes(int B, int N)
{
    if (B < 1 || N < 4 || N > 64) return 0;
    return (int64_t)train_ws(B, N).total;
}

extern "C" int cn_gst_train_step(int B, int N, const float *v_obs, const float *v_pred, const float *loss_mask_rel, const cn_gst_weights *w, const cn_gst_weights *grads,
                                 float p_drop, uint64_t seed, void *workspace, int64_t workspace_bytes, float *loss_out, float *gauss_out, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(B >= 1 && N >= 4 && N <= 64, "cn_gst_train_step: B=%d sequences of N=%d pedestrians outside B >= 1, 4 <= N <= 64 (pad small crowds with absent pedestrians)", B, N);
    CN_REQUIRE(v_obs && v_pred && loss_mask_rel && w && grads && workspace && loss_out, "cn_gst_train_step: null argument");
    CN_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "cn_gst_train_step: p_drop must be in [0, 1)");
    const TrainWs L = train_ws(B, N);
    CN_REQUIRE(workspace_bytes >= (int64_t)L.total && ((uintptr_t)workspace & 15) == 0, "cn_gst_train_step: workspace of %lld bytes needed, got %lld", (long long)L.total,
               (long long)workspace_bytes);
    Wts W;
    Grd G;
    const float *const *wp = reinterpret_cast<const float *const *>(w);
    const float *const *gp = reinterpret_cast<const float *const *>(grads);
    for (int i = 0; i < NPARAM; ++i) {
        CN_REQUIRE(wp[i] && gp[i], "cn_gst_train_step: weight / gradient pointer #%d is null", i);
        W.p[i] = wp[i];
        G.p[i] = const_cast<float *>(gp[i]);
    }
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)workspace;
    float *scratch = (float *)(base + L.scratch), *slabs = (float *)(base + L.slabs), *count = (float *)(base + L.count), *loss_part = (float *)(base + L.loss_part);
    hipLaunchKernelGGL(gst_count_kernel, dim3(1), dim3(NT), 0, st, B, N, loss_mask_rel, count);
    CN_CHECK_LAUNCH();
    hipLaunchKernelGGL(gst_train_kernel, dim3(B), dim3(NT), 0, st, N, v_obs, v_pred, loss_mask_rel, W, p_drop, (unsigned long long)seed, scratch, slabs, count, loss_part, gauss_out);
    CN_CHECK_LAUNCH();
    hipLaunchKernelGGL(gst_grad_reduce_kernel, dim3((NPARAMS + 255) / 256), dim3(256), 0, st, B, slabs, G, loss_part, count, loss_out);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
