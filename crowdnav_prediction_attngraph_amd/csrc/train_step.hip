// train_step.hip -- cn_ppo_minibatch_step: one PPO minibatch (gather -> train-mode forward -> losses -> backward -> parameter gradients)
// as ONE boundary call on gfx950 (include/crowdnav_hip.h).  Reference statements it replaces: rl/networks/storage.py:184-253 (one
// `sample` of recurrent_generator), rl/networks/model.py:82-90 (evaluate_actions), rl/ppo/ppo.py:66-88 (losses, zero_grad, backward).
//
// What is new here (everything else is the existing boundary calls chained on the caller's stream):
//   * tr_gather_kernel / tr_scan_kernel / tr_xlive_kernel: the minibatch is gathered from the rollout storage by env index in one
//     launch (the reference stacks per-env slices in a Python loop; the mirror used a dozen index_select launches), its row offsets
//     and the compacted input rows in two more;
//   * tr_fold_kernel: the affine folds of the network ((q|k|v)_linear o in_proj, out_proj o spatial_linear, Ws^T Wt, (actor.0 ;
//     critic.0) o output_linear -- selfAttn_srnn_temp_node.py:63-91, :160-163, :262-268 compute the unfolded chains) as ONE grouped
//     launch of exact-fp32 64 x 64 tiles, and the chain rule back to the factors as one more; both used to be ~36 launches of
//     cn_small_mm plus autograd's accumulation kernels per optimiser step;
//   * gradients are written straight into the caller's flat bucket (no zero fill, no accumulation pass).
// Summation orders are fixed: the step is deterministic.
#include "common.h"
#include "hh_fused.h"

#include "train_internal.h"

#include <cstdlib>
#include <cstring>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------------------------
// gather of one minibatch (storage.py:209-240): sample b = t * N + j reads env e = env_idx[j] at step t
// ---------------------------------------------------------------------------------------------------------------------------------
struct GatherOut {
    float *rn, *te, *se, *h0, *masks, *act, *vp, *ret, *olp, *adv;
    int *nd;
};

__global__ __launch_bounds__(64) void tr_gather_kernel(cn_ppo_batch b, GatherOut o)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const int t = s / b.N, j = s - t * b.N;
    const int e = b.env_idx[j];
    const size_t te_ = (size_t)t * b.E + e;
    if (lane < 7) o.rn[(size_t)s * 7 + lane] = b.robot_node[te_ * 7 + lane];
    else if (lane < 9) o.te[(size_t)s * 2 + lane - 7] = b.temporal_edges[te_ * 2 + lane - 7];
    else if (lane < 11) o.act[(size_t)s * 2 + lane - 9] = b.actions[te_ * 2 + lane - 9];
    else if (lane == 11) o.masks[s] = b.masks[te_];
    else if (lane == 12) o.vp[s] = b.value_preds[te_];
    else if (lane == 13) o.ret[s] = b.returns[te_];
    else if (lane == 14) o.olp[s] = b.old_logp[te_];
    else if (lane == 15) o.adv[s] = b.adv[te_];
    else if (lane == 16) {
        int nd = (int)b.detected_human_num[te_];
        o.nd[s] = nd < 1 ? 1 : (nd > b.H ? b.H : nd); // 1 .. H rows per sample (crowd_sim_var_num.py:290-292; the kernels assume it)
    }
    const int HD = b.H * b.D;
    const float *src = b.spatial_edges + te_ * HD;
    float *dst = o.se + (size_t)s * HD;
    for (int i = lane; i < HD; i += 64) dst[i] = src[i];
    if (t == 0) { // hidden state of the first step only (storage.py:222)
        const float *hs = b.h0 + (size_t)e * 128;
        float *hd = o.h0 + (size_t)j * 128;
        hd[lane] = hs[lane];
        hd[lane + 64] = hs[lane + 64];
    }
}

// exclusive prefix of nd [n] -> row_off [n + 1], and the compacted input rows x_live [R, D] (the input layer's weight gradient reads them:
// sample s owns rows row_off[s] .. row_off[s + 1] - 1).  Workgroup b owns samples 1024 b .. 1024 b + 1023: it sums everything in front of
// them itself (coalesced, <= 240 KB from the L2 -- cheaper than a second launch or a look-back chain), scans its own 1024 counts, and
// its 16 wavefronts copy the rows of 64 samples each.
__global__ __launch_bounds__(1024) void tr_scan_kernel(int n, int H, int D, const int *__restrict__ nd, int *__restrict__ row_off, const float *__restrict__ se,
                                                       float *__restrict__ x)
{
    __shared__ int part[1024];
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = blockIdx.x * 1024;
    int pre = 0;
    for (int i = tid; i < first; i += 1024) pre += nd[i];
    const int s = first + tid;
    const int mine = s < n ? nd[s] : 0;
    // wave-level inclusive scans (own counts) and sums (prefix), combined through the LDS
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
    if (lane == 63) wsum[wave] = inc;
    if (lane == 0) part[wave] = pre;
    __syncthreads();
    int base = 0, wpre = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { base += part[w]; if (w < wave) wpre += wsum[w]; }
    const int off = base + wpre + inc - mine; // exclusive prefix of sample s
    if (s < n) row_off[s] = off;
    if (s == n - 1) row_off[n] = off + mine;
    __syncthreads(); // everybody has read the wave sums
    part[tid] = off;
    __syncthreads();
    // rows of the wave's 64 samples = ONE contiguous range of x_live: lanes take its floats round robin and find each float's sample by
    // bisection over the wave's 64 offsets in the LDS (independent iterations: a sample-by-sample loop was a chain of 64 load -> store
    // round trips, 40 of the kernel's 52 us)
    const int w0 = wave * 64;
    const int out0 = part[w0];
    const int last = min(first + w0 + 63, n - 1);
    if (first + w0 >= n) return;
    const int out1 = part[last - first] + nd[last];
    for (int idx = out0 * D + lane; idx < out1 * D; idx += 64) {
        const int row = idx / D; // global compacted row
        int lo = 0, hi = last - first - w0; // sample index inside the wave's 64: the last q with part[w0 + q] <= row
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi + 1) >> 1;
            if (part[w0 + mid] <= row) lo = mid; else hi = mid - 1;
        }
        const int sq = first + w0 + lo;
        x[idx] = se[(size_t)sq * H * D + (idx - part[w0 + lo] * D)];
    }
}

__global__ __launch_bounds__(256) void tr_row_totals_kernel(int T, int E, int H, const float *__restrict__ det, int *__restrict__ totals)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    int s = 0;
    for (int t = 0; t < T; ++t) {
        int nd = (int)det[(size_t)t * E + e];
        s += nd < 1 ? 1 : (nd > H ? H : nd);
    }
    totals[e] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// grouped small products: C[M,N] (row stride ldc) = sum over <= 2 segments of A_s[M,K_s] . B_s[K_s,N]  (+ addend[M,N]) (+ add_const)
// in exact fp32 (v_mfma_f32_32x32x2_f32, the arithmetic and k order of cn_small_mm), operands through element strides -- every
// transposed form of the chain rule is a choice of strides, a rank-1 term is a segment with K = 1, a copy is a job without segments.
// One workgroup per 64 x 64 output tile of one job; the table travels as a kernel argument.
// ---------------------------------------------------------------------------------------------------------------------------------
struct FoldSeg {
    const float *A, *B;
    short K, sam, sak, sbk, sbn; // (every dimension and stride of these weight-sized operands is <= 1536)
};
struct FoldJob {
    FoldSeg seg[2];
    const float *addend; // [M,N] with row stride ld_add, or NULL
    float *C;
    int tile0;           // index of this job's first tile in the launch
    float add_const;
    short M, N, nseg, ldc, ld_add;
};
constexpr int FOLD_MAX_JOBS = 36; // 36 x 104 B + 8 B: inside the 4 KB of kernel arguments
static_assert(sizeof(FoldJob) <= 104, "FoldJob grew: the table must stay inside the kernel-argument segment");
struct FoldTable {
    int njobs, ntiles;
    FoldJob job[FOLD_MAX_JOBS];
};

__global__ __launch_bounds__(256) void tr_fold_kernel(const FoldTable tab)
{
    constexpr int T = 64, KT = 32, LS = 36; // tile, K tile, LDS row stride (144 B: conflict-free 16-byte fragment reads)
    __shared__ __attribute__((aligned(16))) float As[T * LS], Bs[T * LS];
    int ji = 0;
    while (ji + 1 < tab.njobs && (int)blockIdx.x >= tab.job[ji + 1].tile0) ++ji;
    const FoldJob &J = tab.job[ji];
    const int M = J.M, N = J.N;
    const int local = (int)blockIdx.x - J.tile0, tn = (N + T - 1) / T;
    const int m0 = (local / tn) * T, n0 = (local % tn) * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int sg = 0; sg < J.nseg; ++sg) {
        const FoldSeg S = J.seg[sg];
        const int K = S.K;
        // this thread's eight elements of each 64 x 32 operand tile; consecutive lanes walk the operand's unit-stride dimension
        int ar[8], ak[8], bn[8], bk[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int i = tid + 256 * p;
            if (S.sak == 1) { ar[p] = i >> 5; ak[p] = i & 31; } else { ar[p] = i & 63; ak[p] = i >> 6; }
            if (S.sbk == 1) { bn[p] = i >> 5; bk[p] = i & 31; } else { bn[p] = i & 63; bk[p] = i >> 6; }
        }
        float ra[8], rb[8];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int m = m0 + ar[p], ka = k0 + ak[p];
                ra[p] = (m < M && ka < K) ? S.A[(long long)m * S.sam + (long long)ka * S.sak] : 0.0f;
                const int n = n0 + bn[p], kb = k0 + bk[p];
                rb[p] = (n < N && kb < K) ? S.B[(long long)kb * S.sbk + (long long)n * S.sbn] : 0.0f;
            }
        };
        fetch(0);
        for (int k0 = 0; k0 < K; k0 += KT) {
            __syncthreads(); // the previous tile is consumed
#pragma unroll
            for (int p = 0; p < 8; ++p) { As[ar[p] * LS + ak[p]] = ra[p]; Bs[bn[p] * LS + bk[p]] = rb[p]; }
            __syncthreads();
            if (k0 + KT < K) fetch(k0 + KT);
#pragma unroll
            for (int g = 0; g < KT / 8; ++g) {
                const f32x4 af = *reinterpret_cast<const f32x4 *>(&As[(wm * 32 + l31) * LS + g * 8 + half * 4]);
                const f32x4 bf = *reinterpret_cast<const f32x4 *>(&Bs[(wn * 32 + l31) * LS + g * 8 + half * 4]);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s4], bf[s4], acc, 0, 0, 0);
            }
        }
    }
    const int col = n0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < M && col < N) {
            float v = acc[r];
            if (J.addend) v += J.addend[(size_t)row * J.ld_add + col];
            J.C[(size_t)row * J.ldc + col] = v + J.add_const;
        }
    }
}

struct FoldBuilder {
    FoldTable t{};
    bool overflow = false;
    FoldJob *add(int M, int N, float *C, int ldc)
    {
        if (t.njobs >= FOLD_MAX_JOBS) { overflow = true; return &t.job[FOLD_MAX_JOBS - 1]; }
        FoldJob &j = t.job[t.njobs++];
        std::memset(&j, 0, sizeof(j));
        j.M = (short)M; j.N = (short)N; j.C = C; j.ldc = (short)ldc; j.tile0 = t.ntiles;
        t.ntiles += ((M + 63) / 64) * ((N + 63) / 64);
        return &j;
    }
    // C = A . B with A[m,k] = A[m * sam + k * sak], B[k,n] = B[k * sbk + n * sbn]
    static void seg(FoldJob *j, int K, const float *A, int sam, int sak, const float *B, int sbk, int sbn)
    {
        FoldSeg &s = j->seg[j->nseg++];
        s.A = A; s.B = B; s.K = (short)K; s.sam = (short)sam; s.sak = (short)sak; s.sbk = (short)sbk; s.sbn = (short)sbn;
    }
    static void plus(FoldJob *j, const float *addend, int ld_add) { j->addend = addend; j->ld_add = (short)ld_add; }
    void copy(int M, int N, float *C, int ldc, const float *src, int ld_src) { plus(add(M, N, C, ldc), src, ld_src); }
    int launch(hipStream_t st)
    {
        CN_REQUIRE(!overflow, "tr_fold: more than %d jobs", FOLD_MAX_JOBS);
        if (!t.ntiles) return CN_OK;
        hipLaunchKernelGGL(tr_fold_kernel, dim3(t.ntiles), dim3(256), 0, st, t);
        CN_CHECK_LAUNCH();
        return CN_OK;
    }
};

// dist_entropy of a DiagGaussian with state-independent logstd (distributions.py:27-29, ppo.py:60): mean over batch and dims of
// 0.5 + 0.5 log(2 pi) + logstd -- the same for every sample
__global__ void tr_entropy_kernel(const float *__restrict__ logstd, float *__restrict__ losses, const float *__restrict__ vl_al, float value_loss_coef,
                                  float *__restrict__ g_losses)
{
    if (threadIdx.x == 0) {
        g_losses[0] = value_loss_coef; // total = value_loss * coef + action_loss - entropy * entropy_coef: the upstream gradients of cn_ppo_loss_bwd
        g_losses[1] = 1.0f;
        losses[0] = vl_al[0];
        losses[1] = vl_al[1];
        losses[2] = 0.5f * ((0.5f + 0.9189385332046727f + logstd[0]) + (0.5f + 0.9189385332046727f + logstd[1]));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// workspace carve-up
// ---------------------------------------------------------------------------------------------------------------------------------
struct Ws {
    // gathered minibatch
    size_t rn, te, se, nd, row_off, h0, masks, act, vp, ret, olp, adv, xlive;
    // folded weights and their gradients
    size_t qkv_w, qkv_b, os_w, os_b, te_w, te_b, ac0_w, ac0_b, d_qkv_w, d_qkv_b, d_os_w, d_os_b, d_te_w, d_te_b, d_ac0_w, d_ac0_b;
    // human-human block: fragment images, saved activations, backward
    size_t frag, e0, x, qkv, attn, out, d_out, d_attn, d_qkv, d_x, d_e0, cls, wT_os, wT_qkv, wT_emb2, part, dbp, e0part, dwb;
    // robot-node sequence
    size_t sv[10], value, logp, d_value, d_logp, rn_fwd, rn_bwd, d_h0;
    // losses
    size_t loss_ws, vl_al, g_losses;
    size_t total;
    int e0_blocks;
};

constexpr int RN_SAVED_W[10] = {256, 384, 256, 0 /* H */, 384, 128, 128, 512, 512, 512}; // cn_rn_saved field widths (attn: H)

Ws carve(int T, int N, int H, int D, int64_t rows)
{
    Ws w{};
    size_t off = 0;
    auto f = [&](size_t n_floats) { size_t o = off; off += ((n_floats * 4 + 255) & ~size_t(255)); return o; }; // 256-byte aligned pieces
    const size_t B = (size_t)T * N, R = (size_t)(rows > 0 ? rows : 1);
    w.rn = f(B * 7); w.te = f(B * 2); w.se = f(B * H * D); w.nd = f(B); w.row_off = f(B + 1); w.h0 = f((size_t)N * 128); w.masks = f(B); w.act = f(B * 2);
    w.vp = f(B); w.ret = f(B); w.olp = f(B); w.adv = f(B); w.xlive = f(R * D);
    w.qkv_w = f(1536 * 512); w.qkv_b = f(1536); w.os_w = f(256 * 512); w.os_b = f(256); w.te_w = f(320 * 256); w.te_b = f(320); w.ac0_w = f(512 * 128); w.ac0_b = f(512);
    w.d_qkv_w = f(1536 * 512); w.d_qkv_b = f(1536); w.d_os_w = f(256 * 512); w.d_os_b = f(256); w.d_te_w = f(320 * 256); w.d_te_b = f(320);
    w.d_ac0_w = f(512 * 128); w.d_ac0_b = f(512);
    w.frag = f((size_t)cn_hh_block_workspace_bytes() / 4);
    w.e0 = f(R * 128); w.x = f(R * 512); w.qkv = f(R * 1536); w.attn = f(R * 512); w.out = f(R * 256);
    w.d_out = f(R * 256); w.d_attn = f(R * 512); w.d_qkv = f(R * 1536); w.d_x = f(R * 512); w.d_e0 = f(R * 128);
    w.cls = f((size_t)cn_hh_attention_workspace_ints((int)B));
    w.wT_os = f(256 * 512); w.wT_qkv = f(1536 * 512); w.wT_emb2 = f(512 * 128); // hi + lo planes of the transposed weights (N * K bf16 each = N * K floats)
    size_t pmax = 0, bmax = 0;
    const int shp[3][2] = {{256, 512}, {1536, 512}, {512, 128}};
    for (auto &q : shp) {
        const size_t sp = (size_t)cn_linear_wgrad_splits((int)R, q[0], q[1]);
        pmax = pmax > sp * q[0] * q[1] ? pmax : sp * q[0] * q[1];
        bmax = bmax > sp * q[0] ? bmax : sp * q[0];
    }
    w.part = f(pmax ? pmax : 1); w.dbp = f(bmax ? bmax : 1);
    w.e0_blocks = (int)(R < 4096 ? R : 4096);
    w.e0part = f((size_t)w.e0_blocks * 128 * (D + 1)); w.dwb = f((size_t)128 * (D + 1));
    for (int i = 0; i < 10; ++i) w.sv[i] = f(B * (size_t)(i == 3 ? H : RN_SAVED_W[i]));
    w.value = f(B); w.logp = f(B); w.d_value = f(B); w.d_logp = f(B);
    w.rn_fwd = f((size_t)cn_rn_seq_fwd_workspace_floats()); w.rn_bwd = f((size_t)cn_rn_seq_workspace_floats(T, N)); w.d_h0 = f((size_t)N * 128);
    w.loss_ws = f((size_t)cn_ppo_loss_workspace_doubles() * 2); w.vl_al = f(4); w.g_losses = f(4);
    w.total = off;
    return w;
}

// Linear (+ ReLU when gate = the layer's output) backward on the bf16x3 kernels, in its two independent halves:
// layer_dx: dX = (dY * [gate > 0]) W   (planes: the transposed split planes of W [N,K], prepared by the step's grouped split launch)
// layer_dw: dW = (dY * [gate > 0])^T X, db = its column sums, written at the given pointers
int layer_dx(int M, int N, int K, const float *dy, const float *gate, const float *planes, float *dx, hipStream_t st)
{
    const uint16_t *hi = reinterpret_cast<const uint16_t *>(planes), *lo = hi + (size_t)N * K;
    return cn_linear_fwd(M, K, N, dy, N, gate, hi, lo, nullptr, 0, dx, K, (void *)st); // dX = dY W as an NT product with W^T [K,N]
}
int layer_dw(int M, int N, int K, const float *dy, const float *gate, const float *inp, float *dW, float *db, char *base, const Ws &L, hipStream_t st)
{
    const int splits = cn_linear_wgrad_splits(M, N, K);
    CN_REQUIRE(splits >= 1, "cn_ppo_minibatch_step: no split-K plan for a %d x %d weight gradient over %d rows", N, K, M);
    return cn_linear_wgrad(M, N, K, dy, N, gate, inp, K, splits, reinterpret_cast<float *>(base + L.part), reinterpret_cast<float *>(base + L.dbp), dW, db, (void *)st);
}

// Library-owned side stream (one per device): carries weight-gradient products that nothing on the critical path waits for, beside the
// dX chain on the caller's stream.  Events are record / wait pairs inside one call; the call joins before it returns its last launches.
struct SideCtx {
    hipStream_t s = nullptr;
    hipEvent_t ev[8] = {};
    bool ok = false;
};
SideCtx *side_ctx()
{
    static SideCtx ctx[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    SideCtx &c = ctx[dev];
    if (!c.ok) {
        if (hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (auto &e : c.ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        c.ok = true;
    }
    return &c;
}

} // namespace

extern "C" int64_t cn_ppo_minibatch_workspace_bytes(int T, int N, int H, int D, int64_t rows)
{
    if (T < 1 || N < 1 || H < 1 || H > 48 || D < 1 || D > 16 || rows < 0) return 0;
    return (int64_t)carve(T, N, H, D, rows).total;
}

extern "C" int cn_ppo_row_totals(int T, int E, int H, const float *detected_human_num, int32_t *totals, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(T >= 1 && E >= 1 && H >= 1 && detected_human_num && totals, "cn_ppo_row_totals: bad argument");
    hipLaunchKernelGGL(tr_row_totals_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, E, H, detected_human_num, totals);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_ppo_minibatch_step(const cn_ppo_batch *bp, int64_t rows, const cn_policy_weights *P, const cn_policy_weights *G, const cn_ppo_hyper *hy,
                                     void *workspace, int64_t workspace_bytes, float *losses_out, float *value_logp_out, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(bp && P && G && hy && workspace && losses_out, "cn_ppo_minibatch_step: null argument");
    const cn_ppo_batch b = *bp;
    CN_REQUIRE(b.T >= 1 && b.N >= 1 && b.E >= b.N && b.H >= 1 && b.H <= 48 && b.D >= 1 && b.D <= 16,
               "cn_ppo_minibatch_step: T=%d N=%d E=%d H=%d D=%d outside T, N >= 1, N <= E, 1 <= H <= 48, 1 <= D <= 16", b.T, b.N, b.E, b.H, b.D);
    CN_REQUIRE(b.env_idx && b.robot_node && b.temporal_edges && b.spatial_edges && b.detected_human_num && b.h0 && b.masks && b.actions && b.value_preds &&
               b.returns && b.old_logp && b.adv, "cn_ppo_minibatch_step: null storage tensor");
    const int64_t Bs = (int64_t)b.T * b.N;
    CN_REQUIRE(rows >= Bs && rows <= Bs * b.H, "cn_ppo_minibatch_step: rows=%lld outside [B, B * H] = [%lld, %lld]", (long long)rows, (long long)Bs, (long long)(Bs * b.H));
    CN_REQUIRE(Bs * b.H < (1LL << 31) / 1536, "cn_ppo_minibatch_step: minibatch too large for 32-bit element offsets");
    {
        const void *const *pp = reinterpret_cast<const void *const *>(P), *const *gp = reinterpret_cast<const void *const *>(G);
        for (size_t i = 0; i < sizeof(cn_policy_weights) / sizeof(void *); ++i)
            CN_REQUIRE(pp[i] && gp[i] && ((uintptr_t)pp[i] & 15) == 0 && ((uintptr_t)gp[i] & 15) == 0,
                       "cn_ppo_minibatch_step: parameter / gradient pointer #%zu is null or not 16-byte aligned", i);
    }
    const Ws L = carve(b.T, b.N, b.H, b.D, rows);
    CN_REQUIRE(workspace_bytes >= (int64_t)L.total && ((uintptr_t)workspace & 255) == 0,
               "cn_ppo_minibatch_step: workspace of %lld bytes (256-byte aligned) needed, got %lld", (long long)L.total, (long long)workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)workspace;
    auto F = [&](size_t off) { return reinterpret_cast<float *>(base + off); };
    auto g = [](const float *p) { return const_cast<float *>(p); }; // the gradient struct reuses cn_policy_weights: its pointers are written
    const int T = b.T, N = b.N, H = b.H, D = b.D, B = (int)Bs, R = (int)rows;
    int *nd = reinterpret_cast<int *>(base + L.nd), *row_off = reinterpret_cast<int *>(base + L.row_off);
    int rc;

    // ---- the minibatch, its row offsets, its compacted input rows ----
    {
        GatherOut o{F(L.rn), F(L.te), F(L.se), F(L.h0), F(L.masks), F(L.act), F(L.vp), F(L.ret), F(L.olp), F(L.adv), nd};
        hipLaunchKernelGGL(tr_gather_kernel, dim3(B), dim3(64), 0, st, b, o);
        CN_CHECK_LAUNCH();
        hipLaunchKernelGGL(tr_scan_kernel, dim3((B + 1023) / 1024), dim3(1024), 0, st, B, H, D, nd, row_off, F(L.se), F(L.xlive));
        CN_CHECK_LAUNCH();
    }

    // ---- affine folds of this step's weights (one grouped launch) ----
    {
        FoldBuilder fb;
        const float *lin_w[3] = {P->q_w, P->k_w, P->v_w}, *lin_b[3] = {P->q_b, P->k_b, P->v_b};
        for (int i = 0; i < 3; ++i) { // (q|k|v)_linear o in_proj: Wc_i = W_in[i] W_i, bc_i = W_in[i] b_i + b_in[i]
            const float *Wi = P->in_proj_w + (size_t)i * 512 * 512;
            FoldBuilder::seg(fb.add(512, 512, F(L.qkv_w) + (size_t)i * 512 * 512, 512), 512, Wi, 512, 1, lin_w[i], 512, 1);
            FoldJob *j = fb.add(512, 1, F(L.qkv_b) + i * 512, 1);
            FoldBuilder::seg(j, 512, Wi, 512, 1, lin_b[i], 1, 0);
            FoldBuilder::plus(j, P->in_proj_b + i * 512, 1);
        }
        { // out_proj o spatial_linear
            FoldBuilder::seg(fb.add(256, 512, F(L.os_w), 512), 512, P->spatial_linear_w, 512, 1, P->out_proj_w, 512, 1);
            FoldJob *j = fb.add(256, 1, F(L.os_b), 1);
            FoldBuilder::seg(j, 512, P->spatial_linear_w, 512, 1, P->out_proj_b, 1, 0);
            FoldBuilder::plus(j, P->spatial_linear_b, 1);
        }
        { // u = Ws^T (Wt r + bt): rows 0..255 of te; rows 256..319 = encoder_linear
            FoldBuilder::seg(fb.add(256, 256, F(L.te_w), 256), 64, P->attn_spatial_w, 1, 256, P->attn_temporal_w, 256, 1);
            FoldBuilder::seg(fb.add(256, 1, F(L.te_b), 1), 64, P->attn_spatial_w, 1, 256, P->attn_temporal_b, 1, 0);
            fb.copy(64, 256, F(L.te_w) + 256 * 256, 256, P->enc_w, 256);
            fb.copy(64, 1, F(L.te_b) + 256, 1, P->enc_b, 1);
        }
        const float *w0[2] = {P->actor0_w, P->critic0_w}, *b0[2] = {P->actor0_b, P->critic0_b};
        for (int i = 0; i < 2; ++i) { // (actor.0 ; critic.0) o output_linear
            FoldBuilder::seg(fb.add(256, 128, F(L.ac0_w) + (size_t)i * 256 * 128, 128), 256, w0[i], 256, 1, P->out_w, 128, 1);
            FoldJob *j = fb.add(256, 1, F(L.ac0_b) + i * 256, 1);
            FoldBuilder::seg(j, 256, w0[i], 256, 1, P->out_b, 1, 0);
            FoldBuilder::plus(j, b0[i], 1);
        }
        if ((rc = fb.launch(st))) return rc;
    }

    // ---- train-mode forward: human-human block (one launch), robot-node sequence ----
    if ((rc = cn_hh_block_fwd(B, H, D, F(L.se), row_off, P->emb0_w, P->emb0_b, P->emb2_w, P->emb2_b, F(L.qkv_w), F(L.qkv_b), F(L.os_w), F(L.os_b), 0.125f,
                              base + L.frag, F(L.e0), F(L.x), F(L.qkv), F(L.attn), F(L.out), stream))) return rc;
    cn_rn_weights rw{P->robot_linear_w, P->robot_linear_b, F(L.te_w), F(L.te_b), P->edge_embed_w, P->edge_embed_b, P->gru_w_ih, P->gru_b_ih, P->gru_w_hh, P->gru_b_hh,
                     F(L.ac0_w), F(L.ac0_b), P->actor2_w, P->actor2_b, P->critic2_w, P->critic2_b, P->critic_linear_w, P->critic_linear_b, P->fc_mean_w, P->fc_mean_b,
                     P->logstd};
    cn_rn_saved sv{F(L.sv[0]), F(L.sv[1]), F(L.sv[2]), F(L.sv[3]), F(L.sv[4]), F(L.sv[5]), F(L.sv[6]), F(L.sv[7]), F(L.sv[8]), F(L.sv[9])};
    { // every split-plane image the step needs (the sequence's forward and backward weights, the transposed weights of the block's dX products): one launch
        CnSplitJob jobs[CN_SPLIT_MAX_JOBS];
        int nj = rn_seq_prep_jobs(&rw, F(L.rn_fwd), F(L.rn_bwd), T, N, jobs);
        jobs[nj++] = cn_split_job(F(L.os_w), 256, 512, 1, 0, F(L.wT_os));
        jobs[nj++] = cn_split_job(F(L.qkv_w), 1536, 512, 1, 0, F(L.wT_qkv));
        jobs[nj++] = cn_split_job(P->emb2_w, 512, 128, 1, 0, F(L.wT_emb2));
        if ((rc = cn_split_group_launch(jobs, nj, st))) return rc;
    }
    if ((rc = rn_seq_fwd_impl(T, N, H, F(L.rn), F(L.te), F(L.out), row_off, F(L.h0), F(L.masks), F(L.act), &rw, &sv, F(L.rn_fwd), F(L.value), F(L.logp), stream, true))) return rc;

    if (value_logp_out) {
        CN_HIP(hipMemcpyAsync(value_logp_out, F(L.value), (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, st));
        CN_HIP(hipMemcpyAsync(value_logp_out + B, F(L.logp), (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, st));
    }

    // ---- losses (ppo.py:66-86) and their gradients w.r.t. values / log-probs ----
    if ((rc = cn_ppo_loss_fwd(B, F(L.value), F(L.logp), F(L.olp), F(L.adv), F(L.vp), F(L.ret), hy->clip_param, hy->use_clipped_value_loss,
                              reinterpret_cast<double *>(base + L.loss_ws), F(L.vl_al), stream))) return rc;
    hipLaunchKernelGGL(tr_entropy_kernel, dim3(1), dim3(64), 0, st, P->logstd, losses_out, F(L.vl_al), hy->value_loss_coef, F(L.g_losses));
    CN_CHECK_LAUNCH();
    if ((rc = cn_ppo_loss_bwd(B, F(L.value), F(L.logp), F(L.olp), F(L.adv), F(L.vp), F(L.ret), hy->clip_param, hy->use_clipped_value_loss, F(L.g_losses),
                              F(L.d_value), F(L.d_logp), stream))) return rc;

    // ---- backward of the robot-node sequence: direct parameters straight into the bucket, folded ones into the workspace ----
    cn_rn_grads rg{g(G->robot_linear_w), g(G->robot_linear_b), F(L.d_te_w), F(L.d_te_b), g(G->edge_embed_w), g(G->edge_embed_b), g(G->gru_w_ih), g(G->gru_b_ih),
                   g(G->gru_w_hh), g(G->gru_b_hh), F(L.d_ac0_w), F(L.d_ac0_b), g(G->actor2_w), g(G->actor2_b), g(G->critic2_w), g(G->critic2_b),
                   g(G->critic_linear_w), g(G->critic_linear_b), g(G->fc_mean_w), g(G->fc_mean_b), g(G->logstd)};
    // The weight-gradient products of the sequence (eight small split-K launches + their reductions, ~0.6 ms at 61 k samples) and the one of
    // out_proj o spatial_linear run on the side stream beside the dX chain, the GRU, the two attention backward kernels (HBM-bound: the
    // matrix cores are idle under them); CN_TRAIN_SIDE_STREAM=0 keeps everything on the caller's stream (A/B timing).
    static const int use_side = getenv("CN_TRAIN_SIDE_STREAM") ? atoi(getenv("CN_TRAIN_SIDE_STREAM")) : 0; // 1: the sequence's products, 2: + out_proj o spatial_linear
    SideCtx *sc = use_side ? side_ctx() : nullptr;
    hipStream_t side = sc ? sc->s : nullptr;
    if (sc) { // nothing of an earlier call may still be running there (a caller that switched streams between calls)
        CN_HIP(hipEventRecord(sc->ev[7], st));
        CN_HIP(hipStreamWaitEvent(side, sc->ev[7], 0));
    }
    float *heads = nullptr; // packed head gradients: fc_mean.w [2,256] | critic_linear.w [256] | fc_mean.b [2] | critic_linear.b [1] | logstd [2]
    if ((rc = rn_seq_bwd_impl(T, N, H, F(L.rn), F(L.te), F(L.out), row_off, F(L.masks), F(L.act), &rw, &sv, F(L.d_value), F(L.d_logp), F(L.rn_bwd), F(L.d_out),
                              F(L.d_h0), &rg, stream, side, sc ? sc->ev : nullptr, true, &heads))) return rc;

    // ---- backward of the human-human block: the per-layer kernels on the saved activations, in reverse order ----
    // out = relu(attn Wos^T + b): d_out is complete since the sequence's attention backward, i.e. before the side stream's last wait
    if ((rc = layer_dw(R, 256, 512, F(L.d_out), F(L.out), F(L.attn), F(L.d_os_w), F(L.d_os_b), base, L, side && use_side >= 2 ? side : st))) return rc;
    if ((rc = layer_dx(R, 256, 512, F(L.d_out), F(L.out), F(L.wT_os), F(L.d_attn), st))) return rc;
    if ((rc = cn_hh_attention_bwd(B, H, F(L.qkv), row_off, F(L.d_attn), 0.125f, F(L.d_qkv), reinterpret_cast<int *>(base + L.cls), 0, stream))) return rc;
    if ((rc = layer_dx(R, 1536, 512, F(L.d_qkv), nullptr, F(L.wT_qkv), F(L.d_x), st))) return rc;                                                   // qkv = x Wc^T + bc
    if (sc) { // join: the next weight gradient reuses the partial-sum buffers, and from here on everything is on the caller's stream again
        CN_HIP(hipEventRecord(sc->ev[6], side));
        CN_HIP(hipStreamWaitEvent(st, sc->ev[6], 0));
    }
    if ((rc = layer_dw(R, 1536, 512, F(L.d_qkv), nullptr, F(L.x), F(L.d_qkv_w), F(L.d_qkv_b), base, L, st))) return rc;
    if ((rc = layer_dx(R, 512, 128, F(L.d_x), F(L.x), F(L.wT_emb2), F(L.d_e0), st))) return rc;                                                     // x = relu(e0 W2^T + b2)
    if ((rc = layer_dw(R, 512, 128, F(L.d_x), F(L.x), F(L.e0), g(G->emb2_w), g(G->emb2_b), base, L, st))) return rc;
    if ((rc = cn_embed0_bwd(R, D, F(L.xlive), F(L.e0), F(L.d_e0), L.e0_blocks, F(L.e0part), F(L.dwb), stream))) return rc;

    // ---- chain rule of the folds back to the factors + the strided pieces (one grouped launch) ----
    {
        FoldBuilder fb;
        const float *lin_w[3] = {P->q_w, P->k_w, P->v_w}, *lin_b[3] = {P->q_b, P->k_b, P->v_b};
        float *glin_w[3] = {g(G->q_w), g(G->k_w), g(G->v_w)}, *glin_b[3] = {g(G->q_b), g(G->k_b), g(G->v_b)};
        for (int i = 0; i < 3; ++i) {
            const float *Wi = P->in_proj_w + (size_t)i * 512 * 512, *dWc = F(L.d_qkv_w) + (size_t)i * 512 * 512, *dbc = F(L.d_qkv_b) + i * 512;
            // d W_in[i] = dWc_i W_i^T + dbc_i (x) b_i ;  d W_i = W_in[i]^T dWc_i ;  d b_i = W_in[i]^T dbc_i
            FoldJob *j = fb.add(512, 512, g(G->in_proj_w) + (size_t)i * 512 * 512, 512);
            FoldBuilder::seg(j, 512, dWc, 512, 1, lin_w[i], 1, 512);
            FoldBuilder::seg(j, 1, dbc, 1, 0, lin_b[i], 0, 1);
            FoldBuilder::seg(fb.add(512, 512, glin_w[i], 512), 512, Wi, 1, 512, dWc, 512, 1);
            FoldBuilder::seg(fb.add(512, 1, glin_b[i], 1), 512, Wi, 1, 512, dbc, 1, 0);
        }
        fb.copy(1536, 1, g(G->in_proj_b), 1, F(L.d_qkv_b), 1);
        { // os = spatial_linear o out_proj
            FoldJob *j = fb.add(256, 512, g(G->spatial_linear_w), 512);
            FoldBuilder::seg(j, 512, F(L.d_os_w), 512, 1, P->out_proj_w, 1, 512);
            FoldBuilder::seg(j, 1, F(L.d_os_b), 1, 0, P->out_proj_b, 0, 1);
            fb.copy(256, 1, g(G->spatial_linear_b), 1, F(L.d_os_b), 1);
            FoldBuilder::seg(fb.add(512, 512, g(G->out_proj_w), 512), 256, P->spatial_linear_w, 1, 512, F(L.d_os_w), 512, 1);
            FoldBuilder::seg(fb.add(512, 1, g(G->out_proj_b), 1), 256, P->spatial_linear_w, 1, 512, F(L.d_os_b), 1, 0);
        }
        { // te rows 0..255: U = Ws^T Wt, ub = Ws^T bt.  d Ws = Wt dU^T + bt (x) dub ;  d Wt = Ws dU ;  d bt = Ws dub ;  d bs = 0 (the softmax cannot see it)
            FoldJob *j = fb.add(64, 256, g(G->attn_spatial_w), 256);
            FoldBuilder::seg(j, 256, P->attn_temporal_w, 256, 1, F(L.d_te_w), 1, 256);
            FoldBuilder::seg(j, 1, P->attn_temporal_b, 1, 0, F(L.d_te_b), 0, 1);
            FoldBuilder::seg(fb.add(64, 256, g(G->attn_temporal_w), 256), 256, P->attn_spatial_w, 256, 1, F(L.d_te_w), 256, 1);
            FoldBuilder::seg(fb.add(64, 1, g(G->attn_temporal_b), 1), 256, P->attn_spatial_w, 256, 1, F(L.d_te_b), 1, 0);
            fb.add(64, 1, g(G->attn_spatial_b), 1); // exact zero
            fb.copy(64, 256, g(G->enc_w), 256, F(L.d_te_w) + 256 * 256, 256);
            fb.copy(64, 1, g(G->enc_b), 1, F(L.d_te_b) + 256, 1);
        }
        { // ac0 = (actor.0 ; critic.0) o output_linear
            const float *w0[2] = {P->actor0_w, P->critic0_w};
            float *gw0[2] = {g(G->actor0_w), g(G->critic0_w)}, *gb0[2] = {g(G->actor0_b), g(G->critic0_b)};
            for (int i = 0; i < 2; ++i) {
                const float *dA = F(L.d_ac0_w) + (size_t)i * 256 * 128, *db = F(L.d_ac0_b) + i * 256;
                FoldJob *j = fb.add(256, 256, gw0[i], 256); // d W0_i = dA_i Wo^T + db_i (x) bo
                FoldBuilder::seg(j, 128, dA, 128, 1, P->out_w, 1, 128);
                FoldBuilder::seg(j, 1, db, 1, 0, P->out_b, 0, 1);
                fb.copy(256, 1, gb0[i], 1, db, 1);
            }
            FoldJob *j = fb.add(256, 128, g(G->out_w), 128); // d Wo = actor.0^T dA_actor + critic.0^T dA_critic
            FoldBuilder::seg(j, 256, w0[0], 1, 256, F(L.d_ac0_w), 128, 1);
            FoldBuilder::seg(j, 256, w0[1], 1, 256, F(L.d_ac0_w) + 256 * 128, 128, 1);
            j = fb.add(256, 1, g(G->out_b), 1);
            FoldBuilder::seg(j, 256, w0[0], 1, 256, F(L.d_ac0_b), 1, 0);
            FoldBuilder::seg(j, 256, w0[1], 1, 256, F(L.d_ac0_b) + 256, 1, 0);
        }
        // input layer: dwb [128, D + 1] = per output column the D weight gradients, then the bias gradient
        fb.copy(128, D, g(G->emb0_w), D, F(L.dwb), D + 1);
        fb.copy(128, 1, g(G->emb0_b), 1, F(L.dwb) + D, D + 1);
        // heads: scattered from the packed reduction; d(-coef * entropy) / d logstd_k = -coef / 2 on top of the log-prob path's gradient
        fb.copy(2, 256, g(G->fc_mean_w), 256, heads, 256);
        fb.copy(1, 256, g(G->critic_linear_w), 256, heads + 512, 256);
        fb.copy(2, 1, g(G->fc_mean_b), 1, heads + 768, 1);
        fb.copy(1, 1, g(G->critic_linear_b), 1, heads + 770, 1);
        fb.add(2, 1, g(G->logstd), 1);
        FoldBuilder::plus(&fb.t.job[fb.t.njobs - 1], heads + 771, 1);
        fb.t.job[fb.t.njobs - 1].add_const = -0.5f * hy->entropy_coef;
        if ((rc = fb.launch(st))) return rc;
    }
    return CN_OK;
}
