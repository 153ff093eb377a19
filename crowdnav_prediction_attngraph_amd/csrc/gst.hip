// gst.hip -- Gumbel Social Transformer trajectory predictor (inference) and the VecPretextNormalize post-processing on
// gfx950, batched over all envs.  Reference (under the reference repo):
//   gst_updated/scripts/wrapper/crowd_nav_interface_parallel.py:45-114   mask / displacement preparation, cumulative outputs
//   gst_updated/src/gumbel_social_transformer/st_model.py:271-455         st_model.forward ('faster_lstm', 'recursive' decode)
//   .../gumbel_social_transformer.py:43-96, node_encoder_layer_no_ghost.py:25-66, mha.py:236-242 (spatial_num_heads_edges = 0:
//       fully connected; float adjacency mask multiplied AFTER the softmax, then renormalised)
//   rl/vec_env/vec_pretext_normalize.py:85-191                            history buffers, social penalty, edge write, sort
// Shipped hyper-parameters only: embedding 64, 8 heads x 8, 1 encoder layer, FFN 128, LSTM 64, obs 5 / pred 5, output 5.
//
// Row index space: a "slab" has S time slices (S = 5 for the observation pass, 1 for each decode step);
// row = (env * S + t) * H + human.  Round 2: a forward is 16 launches instead of ~66 -- the NodeEncoderLayer is ONE kernel
// (gst_layer_kernel: activations of a tile of whole groups in LDS, bf16x3 MFMA), the LSTM over a slab is ONE kernel (gst_lstm_kernel:
// weights in registers, h / c on chip across the slices); the head and the wrapper kernels are wave-per-row.  All work is enqueued on
// the caller's stream.
#include "common.h"
#include "rn_fused.h" // rn_fused_bake: bf16 hi/lo MFMA-fragment image of a weight matrix
#include <utility>

#include <cmath>
#include <cstdlib>
#include <new>

namespace {

constexpr int GT = 5, GP = 5; // obs_seq_len, pred_seq_len
constexpr float GST_INVALID = -999.0f;

// masks + displacements (crowd_nav_interface_parallel.py:71-89).  traj element (e,h,t) lives at
// traj[e*se + h*sh + ((rot + t*step) % ring)*st] (+0/+1 for x/y); mask likewise with strides me/mh/mt.  (A plain [.., GT, ..] input: rot 0,
// step 1, ring GT; the wrapper's history ring of (GT-1)*I + 1 slots read every I-th, oldest first: vec_pretext_normalize.py:56-57, :133-134.)
// NOTE loss_mask_rel_obs[t>=1] = mask[t-1] * mask[T-1] exactly as written in the reference (:76).
__global__ __launch_bounds__(256) void gst_obs_prep_kernel(int E, int H, const float *__restrict__ traj, long long se, long long sh, long long st,
                                                           const uint8_t *__restrict__ mask_u8, const float *__restrict__ mask_f,
                                                           long long me, long long mh, long long mt, int rot, int step, int ring, float *__restrict__ m_rel,
                                                           float *__restrict__ lm_fp, float *__restrict__ rel, float *__restrict__ last_pos)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * H) return;
    const int e = idx / H, h = idx - e * H;
    float m[GT], px[GT], py[GT];
#pragma unroll
    for (int t = 0; t < GT; ++t) {
        const int tt = (rot + t * step) % ring;
        const size_t mo = (size_t)e * me + (size_t)h * mh + (size_t)tt * mt;
        m[t] = mask_u8 ? (mask_u8[mo] ? 1.0f : 0.0f) : mask_f[mo];
        const size_t to = (size_t)e * se + (size_t)h * sh + (size_t)tt * st;
        px[t] = traj[to]; py[t] = traj[to + 1];
    }
#pragma unroll
    for (int t = 0; t < GT; ++t) {
        const float mr = t == 0 ? m[0] : m[t - 1] * m[GT - 1];
        const float dx = t == 0 ? 0.0f : px[t] - px[t - 1], dy = t == 0 ? 0.0f : py[t] - py[t - 1];
        const size_t r = ((size_t)e * GT + t) * H + h;
        m_rel[r] = mr;
        rel[2 * r] = GST_INVALID * (1.0f - mr) + dx * mr;
        rel[2 * r + 1] = GST_INVALID * (1.0f - mr) + dy * mr;
        if (t == GT - 1) lm_fp[idx] = mr;
    }
    last_pos[2 * idx] = px[GT - 1]; last_pos[2 * idx + 1] = py[GT - 1];
}

// ------------------------------------------------------------------------------------------------------------------
// The whole NodeEncoderLayer as ONE kernel (round 2): embedding -> LayerNorm * mask -> in_proj -> 8-head attention (multiplicative mask
// + renormalisation) -> out_proj + residual -> LayerNorm -> FFN 64-128-64 + residual -> xs.
// As eight launches (three of them K = 64 GEMMs that move 1-3 KB per row through HBM for 25-50 kFLOP) this chain was 2/3 of the
// GST step; here a workgroup (4 wavefronts) owns a tile of TG whole groups (<= 80 rows), keeps every activation of the tile in LDS
// (108 KB; 8 wavefronts) and streams the 196 KB weight image (bf16 hi/lo MFMA fragments, the bake of rn_fused.hip) from L2 once per tile.  Products
// run in bf16x3 split precision on v_mfma_f32_16x16x32_bf16, "transposed" (A = weights, B = activations): the C layout is 4
// consecutive output features of one row per lane = a float4 store into the next stage's [row][feature] image.
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef CN_GL_RT
#define CN_GL_RT 5
#endif
#ifndef CN_GL_WAVES
#define CN_GL_WAVES 8   // wavefronts per workgroup of the encoder-layer kernel: 8 = two teams of four on alternate row tiles (one workgroup per CU);
#endif                  // 4 = one team, and with CN_GL_RT = 3 (48 rows, 64 KB of LDS) TWO independent workgroups per CU whose phases drift apart
#ifndef CN_GST_WGS
#define CN_GST_WGS 1   // workgroups per CU the LSTM kernel is sized for
#endif
constexpr int GL_NW = CN_GL_WAVES, GL_TEAMS = GL_NW / 4, GL_THREADS = 64 * GL_NW, GL_WGS = GL_NW == 4 ? 2 : 1;
#ifndef CN_LS_ROWS
#define CN_LS_ROWS 0   // nodes per LSTM tile: 0 = chosen per launch between 32 and 80 (gst_lstm); 32 / 80 force one
#endif
constexpr int GL_RT = CN_GL_RT, GL_ROWS = 16 * GL_RT;            // row tiles / rows per workgroup tile
constexpr int GL_SA = 68, GL_SQ = 196;                    // LDS row strides (floats): 16-byte aligned, lanes of a float4 read on distinct banks
constexpr int GL_OA = 0, GL_OQ = GL_OA + GL_ROWS * GL_SA, GL_OT = GL_OQ + GL_ROWS * GL_SQ, GL_OM = GL_OT + GL_ROWS * GL_SA;
constexpr int GL_OX = GL_OM + GL_ROWS, GL_OC = GL_OX + 2 * GL_ROWS; // staged (x, y) inputs; constants of the embedding's LayerNorm
constexpr int GL_OB = GL_OC + 16;            // biases: in_proj 192 | out_proj 64 | linear1 128 | linear2 64 (read by every tile: staged once)
constexpr int GL_ORM = GL_OB + 448;         // row map of a tile of listed groups (ints): row r of the tile -> row of x2 / mask / xs
constexpr int GL_LDS_FLOATS = GL_ORM + GL_ROWS;

struct GstLayerArgs {
    const float *x2, *mask;                                   // [rows,2], [rows]
    const float *emb_w, *emb_b, *n_w, *n_b, *n1_w, *n1_b;     // fp32 small tensors
    const float *f_in, *in_b, *f_out, *out_b, *f_l1, *l1_b, *f_l2, *l2_b; // baked fragments + biases
    float *xs;                                                // [rows,64] encoded rows (input of the LSTM)
    const int *glist, *gcount; int gbase;                     // optional: only the groups glist[0 .. gbase + *gcount) (gst_reuse_kernel), any order
};

// The weight fragments of a stage for the feature blocks fb_first, fb_first + fb_step, ... of this wavefront.  Wfrag: [fb][K/32][hi,lo]
// [64 lanes][8 bf16].  Loaded one phase AHEAD of their use (the kernel is a chain of short dependent phases at one workgroup per CU:
// fetched at the start of its own stage, every stage began with a cold L2 round trip -- 90 of them per workgroup and step).
template <int K, int NFB>
struct GlW { bf16x8 h[K / 32][NFB], l[K / 32][NFB]; };

template <int K, int NFB>
__device__ __forceinline__ void gl_load(GlW<K, NFB> &w, const float *__restrict__ Wfrag_, int fb_first, int fb_step, int lane)
{
    const bf16x8 *__restrict__ Wfrag = reinterpret_cast<const bf16x8 *>(Wfrag_);
#pragma unroll
    for (int ks = 0; ks < K / 32; ++ks)
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
            const size_t base = ((size_t)(fb_first + j * fb_step) * (K / 32) + ks) * 128 + lane;
            w.h[ks][j] = Wfrag[base]; w.l[ks][j] = Wfrag[base + 64];
        }
}

// out[row][out_off + f] = act(W[f] . in[row] + bias[f] (+ res[row][f])) for this wavefront's feature blocks and all row tiles.
// `out` may be LDS or global.
template <int K, int NFB, bool RELU>
__device__ __forceinline__ void gl_stage(const GlW<K, NFB> &w, int fb_first, int fb_step, const float *__restrict__ bias, const float *in,
                                         int in_stride, float *out, size_t out_stride, int out_off, const float *res, int res_stride, int nrows, int lane,
                                         int rt_first, const int *rowmap = nullptr)
{
    const int i = lane & 15, g = lane >> 4;
    constexpr int KS = K / 32;
    // the biases are read BEFORE the row-tile loop: inside it every load of them would have to wait for the previous tile's stores
    // (the compiler cannot prove that `out` does not alias them) -- one global round trip per row tile and stage
    f32x4 bv[NFB];
#pragma unroll
    for (int j = 0; j < NFB; ++j) bv[j] = bias ? *reinterpret_cast<const f32x4 *>(bias + (fb_first + j * fb_step) * 16 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    // the workgroup's two wavefront teams (4 wavefronts each, same feature blocks) take the even / the odd row tiles
#pragma unroll
    for (int rt2 = 0; rt2 < (GL_RT + GL_TEAMS - 1) / GL_TEAMS; ++rt2) {
        const int rt = GL_TEAMS * rt2 + rt_first;
        if (16 * rt >= nrows) break; // wave-uniform
        f32x4 acc[NFB];
#pragma unroll
        for (int j = 0; j < NFB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *inr = in + (16 * rt + i) * in_stride + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(inr + 32 * ks), x1 = *reinterpret_cast<const f32x4 *>(inr + 32 * ks + 4);
            bf16x8 bh, bl;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const __bf16 h0 = (__bf16)x0[u], h1 = (__bf16)x1[u];
                bh[u] = h0; bh[4 + u] = h1;
                bl[u] = (__bf16)(x0[u] - (float)h0); bl[4 + u] = (__bf16)(x1[u] - (float)h1);
            }
#pragma unroll
            for (int j = 0; j < NFB; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.l[ks][j], bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[ks][j], bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[ks][j], bh, acc[j], 0, 0, 0);
            }
        }
        const int row = 16 * rt + i;
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
            const int f0 = (fb_first + j * fb_step) * 16 + 4 * g;
            f32x4 v = acc[j] + bv[j];
            if (res) v += *reinterpret_cast<const f32x4 *>(res + row * res_stride + f0);
            if (RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (row < nrows) *reinterpret_cast<f32x4 *>(out + (size_t)(rowmap ? rowmap[row] : row) * out_stride + out_off + f0) = v;
        }
    }
}

#ifdef GST_TIMING
__device__ long long *g_gst_tim = nullptr; // [block][16] phase cycle sums of thread 0 (measurement build only): slots 0..7 the layer kernel's phases, 8..12 the LSTM kernel's
#define GL_T(k) do { if (g_gst_tim && threadIdx.x == 0) { const long long now_ = clock64(); g_gst_tim[(size_t)blockIdx.x * 16 + (k)] += now_ - tlast_; tlast_ = now_; } } while (0)
#else
#define GL_T(k) do {} while (0)
#endif

__global__ __launch_bounds__(GL_THREADS, 2) void gst_layer_kernel(int rows, int H, int TG, GstLayerArgs a)
{
#ifdef GST_TIMING
    long long tlast_ = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float *A0 = lds_f + GL_OA, *QKV = lds_f + GL_OQ, *ATT = lds_f + GL_OT, *MSK = lds_f + GL_OM, *XY = lds_f + GL_OX, *EC = lds_f + GL_OC;
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, team = wave8 >> 2; // feature-block owner; row-tile parity (8 wavefronts: two per SIMD hide each other's latencies)
    const int tile_rows = TG * H;
    int *RM = reinterpret_cast<int *>(lds_f + GL_ORM);
    if (a.glist) rows = (a.gbase + *a.gcount) * H; // the listed groups only (the others' rows were taken over from the previous call: gst_reuse_kernel)
    const int n_tiles = (rows + tile_rows - 1) / tile_rows;
    // LayerNorm of the embedding v_f = We[f][0] x + We[f][1] y + be[f] needs no reduction per row: its mean and variance over the 64
    // features are a linear / quadratic form of (x, y) with nine constants of the weight snapshot (computed here once per workgroup)
    if (wave8 == 0) {
        const float w0 = a.emb_w[2 * lane], w1 = a.emb_w[2 * lane + 1], b0 = a.emb_b[lane];
        const float m0 = wv_sum(w0) * (1.0f / 64.0f), m1 = wv_sum(w1) * (1.0f / 64.0f), mb = wv_sum(b0) * (1.0f / 64.0f);
        const float d0 = w0 - m0, d1 = w1 - m1, db = b0 - mb;
        const float c00 = wv_sum(d0 * d0) * (1.0f / 64.0f), c11 = wv_sum(d1 * d1) * (1.0f / 64.0f), cbb = wv_sum(db * db) * (1.0f / 64.0f);
        const float c01 = wv_sum(d0 * d1) * (1.0f / 64.0f), c0b = wv_sum(d0 * db) * (1.0f / 64.0f), c1b = wv_sum(d1 * db) * (1.0f / 64.0f);
        if (lane == 0) { EC[0] = m0; EC[1] = m1; EC[2] = mb; EC[3] = c00; EC[4] = c11; EC[5] = cbb; EC[6] = c01; EC[7] = c0b; EC[8] = c1b; }
    }
    float *BI = lds_f + GL_OB;
    for (int t = tid; t < 448; t += GL_THREADS) BI[t] = t < 192 ? a.in_b[t] : (t < 256 ? a.out_b[t - 192] : (t < 384 ? a.l1_b[t - 256] : a.l2_b[t - 384]));
    const float ew0 = a.emb_w[2 * lane], ew1 = a.emb_w[2 * lane + 1], eb = a.emb_b[lane], ng = a.n_w[lane], nb = a.n_b[lane];
    const float n1g = a.n1_w[lane], n1b = a.n1_b[lane];
    GlW<64, 3> w_in;
    gl_load<64, 3>(w_in, a.f_in, wave, 4, lane);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int r0 = tile * tile_rows;
        const int nrows = min(tile_rows, rows - r0);   // whole groups (rows is a multiple of H)
        const int nrows16 = (nrows + 15) & ~15;
        // ---- stage the tile's inputs (coalesced), then node_embedding (2 -> 64) + LayerNorm(norm_node) * mask -> A0 (padding rows = 0) ----
        for (int r = tid; r < nrows16; r += GL_THREADS) {
            const bool live = r < nrows;
            size_t src = (size_t)(r0 + r);
            if (a.glist) { // row r of this tile = node r % H of listed group tile * TG + r / H
                const int k = r / H;
                src = live ? (size_t)a.glist[tile * TG + k] * H + (r - k * H) : 0;
                RM[r] = (int)src;
            }
            XY[2 * r] = live ? a.x2[2 * src] : 0.0f;
            XY[2 * r + 1] = live ? a.x2[2 * src + 1] : 0.0f;
            MSK[r] = live ? a.mask[src] : 0.0f;
        }
        __syncthreads();
        for (int r = wave8; r < nrows16; r += GL_NW) {
            const float x = XY[2 * r], y = XY[2 * r + 1], m = MSK[r];
            const float mean = EC[0] * x + EC[1] * y + EC[2];
            const float var = EC[3] * x * x + EC[4] * y * y + EC[5] + 2.0f * (EC[6] * x * y + EC[7] * x + EC[8] * y);
            const float v = ew0 * x + ew1 * y + eb;
            A0[r * GL_SA + lane] = ((v - mean) * rsqrtf(fmaxf(var, 0.0f) + 1e-5f) * ng + nb) * m;
        }
        GlW<64, 1> w_out;
        gl_load<64, 1>(w_out, a.f_out, wave, 4, lane);
        __syncthreads();
        GL_T(0);
        // ---- in_proj: [q | k | v] = W_in x0 + b_in -> QKV ----
        gl_stage<64, 3, false>(w_in, wave, 4, BI, A0, GL_SA, QKV, GL_SQ, 0, nullptr, 0, nrows16, lane, team);
        __syncthreads();
        GL_T(1);
        GlW<64, 2> w_l1;
        gl_load<64, 2>(w_l1, a.f_l1, wave, 4, lane);
        // ---- attention core per group (mha.py:236-242): wavefront w takes groups w, w + 4, ...; lanes enumerate (node, head) ----
        {
            const float scale = 0.35355339059327373f; // 8^-0.5
            for (int pq = tid; pq < nrows * 8; pq += GL_THREADS) { // (row, head) pairs of the whole tile over all threads
                const int row = pq >> 3, hd = pq & 7;
                const int gq = row / H;
                const float *qb = QKV + gq * H * GL_SQ;
                const float *ms = MSK + gq * H;
                const f32x4 qa = *reinterpret_cast<const f32x4 *>(QKV + row * GL_SQ + hd * 8) * scale;
                const f32x4 qc = *reinterpret_cast<const f32x4 *>(QKV + row * GL_SQ + hd * 8 + 4) * scale;
                float mx = -INFINITY;
                float Z = 0.0f, Zm = 0.0f;
                f32x4 aa = f32x4{0.f, 0.f, 0.f, 0.f}, ac = aa;
                if (H <= 20) {
                    // crowds of <= 20 (every BASELINE config with GST in the loop): the scores of the pair stay in registers between the two
                    // passes instead of being computed twice (a third of the core's instructions; same values, same results)
                    float scv[20];
#pragma unroll
                    for (int j = 0; j < 20; ++j) {
                        if (j < H) {
                            const f32x4 ka = *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 64 + hd * 8), kc = *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 68 + hd * 8);
                            float sc = 0.0f;
                            sc += qa[0] * ka[0]; sc += qa[1] * ka[1]; sc += qa[2] * ka[2]; sc += qa[3] * ka[3];
                            sc += qc[0] * kc[0]; sc += qc[1] * kc[1]; sc += qc[2] * kc[2]; sc += qc[3] * kc[3];
                            scv[j] = sc;
                            mx = fmaxf(mx, sc);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 20; ++j) {
                        if (j < H) {
                            const float ex = __builtin_amdgcn_exp2f((scv[j] - mx) * 1.44269504088896340736f); // (v_exp_f32: the argument is <= 0)
                            Z += ex;
                            const float em = ex * ms[j];
                            Zm += em;
                            aa += em * *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 128 + hd * 8);
                            ac += em * *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 132 + hd * 8);
                        }
                    }
                } else {
#pragma unroll 4
                    for (int j = 0; j < H; ++j) {
                        const f32x4 ka = *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 64 + hd * 8), kc = *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 68 + hd * 8);
                        float sc = 0.0f;
                        sc += qa[0] * ka[0]; sc += qa[1] * ka[1]; sc += qa[2] * ka[2]; sc += qa[3] * ka[3];
                        sc += qc[0] * kc[0]; sc += qc[1] * kc[1]; sc += qc[2] * kc[2]; sc += qc[3] * kc[3];
                        mx = fmaxf(mx, sc);
                    }
#pragma unroll 4
                    for (int j = 0; j < H; ++j) {
                        const f32x4 ka = *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 64 + hd * 8), kc = *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 68 + hd * 8);
                        float sc = 0.0f;
                        sc += qa[0] * ka[0]; sc += qa[1] * ka[1]; sc += qa[2] * ka[2]; sc += qa[3] * ka[3];
                        sc += qc[0] * kc[0]; sc += qc[1] * kc[1]; sc += qc[2] * kc[2]; sc += qc[3] * kc[3];
                        const float ex = expf(sc - mx);
                        Z += ex;
                        const float em = ex * ms[j];
                        Zm += em;
                        aa += em * *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 128 + hd * 8);
                        ac += em * *reinterpret_cast<const f32x4 *>(qb + j * GL_SQ + 132 + hd * 8);
                    }
                }
                const float mi = MSK[row];
                const float denom = mi * Zm / Z + 1e-10f;
                const float f = mi / (Z * denom);
                float *o = ATT + row * GL_SA + hd * 8;
                *reinterpret_cast<f32x4 *>(o) = aa * f;
                *reinterpret_cast<f32x4 *>(o + 4) = ac * f;
            }
        }
        // padding rows of ATT (read by the next product's fragments) must be finite
        for (int r = nrows + wave8; r < nrows16; r += GL_NW) ATT[r * GL_SA + lane] = 0.0f;
        __syncthreads();
        GL_T(2);
        // ---- out_proj + residual: x1 = x0 + W_out att + b_out, in place over x0 ----
        gl_stage<64, 1, false>(w_out, wave, 4, BI + 192, ATT, GL_SA, A0, GL_SA, 0, A0, GL_SA, nrows16, lane, team);
        __syncthreads();
        GL_T(3);
        GlW<128, 1> w_l2;
        gl_load<128, 1>(w_l2, a.f_l2, wave, 4, lane);
        // ---- LayerNorm(norm1_node): x2 -> ATT (the attention output is dead) ----
        for (int r = wave8; r < nrows16; r += GL_NW) {
            const float v = A0[r * GL_SA + lane];
            const float mean = wv_sum(v) * (1.0f / 64.0f);
            const float d = v - mean;
            const float var = wv_sum(d * d) * (1.0f / 64.0f);
            ATT[r * GL_SA + lane] = d * rsqrtf(var + 1e-5f) * n1g + n1b;
        }
        __syncthreads();
        GL_T(4);
        // ---- FFN: ff = relu(W_1 x2 + b_1) -> QKV region (q|k|v are dead); xs = x1 + W_2 ff + b_2, in place over x1 ----
        gl_stage<64, 2, true>(w_l1, wave, 4, BI + 256, ATT, GL_SA, QKV, GL_SQ, 0, nullptr, 0, nrows16, lane, team);
        __syncthreads();
        GL_T(5);
        // xs = x1 + W_2 ff + b_2 -> HBM (64 floats per row; the LSTM kernel applies W_ih itself: its [rows, 256] image would be 4x the bytes)
        if (a.glist) gl_stage<128, 1, false>(w_l2, wave, 4, BI + 384, QKV, GL_SQ, a.xs, 64, 0, A0, GL_SA, nrows, lane, team, RM);
        else gl_stage<128, 1, false>(w_l2, wave, 4, BI + 384, QKV, GL_SQ, a.xs + (size_t)r0 * 64, 64, 0, A0, GL_SA, nrows, lane, team);
        if (tile + (int)gridDim.x < n_tiles) gl_load<64, 3>(w_in, a.f_in, wave, 4, lane); // (w_in's registers were free since the in_proj stage)
        __syncthreads(); // the next tile overwrites A0 / MSK
        GL_T(7);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The LSTM over S time slices as ONE kernel: gates = [W_ih | W_hh] [m x_t ; h] + b on the matrix cores (bf16x3), the cell's pointwise
// part, the decode-step blend / the post mask -- h and c never leave the chip between the S steps.  A workgroup owns 64 nodes; its
// 256 x 128 weight image stays in registers (128 VGPRs per wavefront) for all its tiles and steps.  Replaces, per step, a K = 64 GEMM
// whose [N, 256] result went through HBM, the pointwise kernel, and the [rows, 256] input projection of the whole slab.
// ------------------------------------------------------------------------------------------------------------------
// nodes per workgroup tile LS_ROWS (template parameter: 32 or 80, picked per launch by which spreads the batch's tiles better over the
// workgroups -- see gst_lstm); pointwise part: LS_PT nodes per thread
constexpr int LS_SX = 132, LS_SG = 260;
constexpr int ls_lds_floats(int rows) { return rows * LS_SX + rows * LS_SG + 256; }

struct GstLstmArgs {
    const float *xs, *in_mask;      // [E*S*H, 64] encoded rows, [E*S*H] input mask
    const float *f_w, *bih, *bhh;   // baked [256,128] = [W_ih | W_hh]; biases
    float *h, *c;                   // [E*H, 64] state, updated in place
    const float *blend_mask, *post_mask; // [E*H] or null
    // decode head of the step that follows this LSTM pass (hidden2pos + raw2gaussian + the running sums of
    // crowd_nav_interface_parallel.py:99-113), tt = decode step index; head_w == null: no head
    int tt;
    int zero_state;                 // 1: h = c = 0 on entry (the observation pass) instead of two memsets of the state in front of the launch
    const float *head_w, *head_b, *lm_fp, *last_pos;
    float *acc, *out_traj, *x_sample;
};

// tanh / sigmoid on v_exp_f32 (|err| < 3e-7; e^{2x} overflowing to inf gives 1, underflowing to 0 gives -1)
__device__ __forceinline__ float gl_tanh(float x) { return 1.0f - 2.0f / (1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681472f)); }
__device__ __forceinline__ float gl_sigmoid(float x) { return 1.0f / (1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)); }

template <int LS_ROWS>
__global__ __launch_bounds__(512, 2 * CN_GST_WGS) void gst_lstm_kernel(int E, int H, int S, GstLstmArgs a)
{
    constexpr int LS_PT = LS_ROWS / 8, LS_OX = 0, LS_OG = LS_OX + LS_ROWS * LS_SX, LS_OB = LS_OG + LS_ROWS * LS_SG;
#ifdef GST_TIMING
    long long tlast_ = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float *X = lds_f + LS_OX, *G = lds_f + LS_OG, *B = lds_f + LS_OB;
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = E * H, n_tiles = (N + LS_ROWS - 1) / LS_ROWS;
    const int d = tid & 63, q = tid >> 6; // pointwise mapping: hidden unit d of the nodes q, q + 8, ... (8 per thread)
    if (tid < 256) B[tid] = a.bih[tid] + a.bhh[tid];
    // wavefront w owns the feature blocks w and w + 8 of the 16 (64 VGPRs of weights: the whole 256 x 128 image in the registers of the
    // workgroup, never re-read; four blocks per wavefront did not fit beside the rest and spilled into the MFMA loop)
    GlW<128, 2> w;
    gl_load<128, 2>(w, a.f_w, wave8, 8, lane);
    const int i = lane & 15, g = lane >> 4;
    // the head's weights (this thread's hidden unit d) and biases: constants of the launch, fetched once -- and what the head's scalar tail reads
    // per node (running sums, last position, mask) is requested at the START of a tile: fetched behind the reductions, those loads were a chain
    // of two global round trips at the end of every tile and launch (27 % of the kernel's time on its own clock)
    float hw_[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, hb_[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.head_w) {
#pragma unroll
        for (int k = 0; k < 5; ++k) { hw_[k] = a.head_w[64 * k + d]; hb_[k] = a.head_b[k]; }
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int n0 = tile * LS_ROWS;
        float creg[LS_PT], xn[LS_PT], hreg[LS_PT];
        float t_lm = 0.f, t_lp0 = 0.f, t_lp1 = 0.f, t_acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        {
            const int nh = n0 + q + 8 * d; // the node whose scalar tail this thread runs (lanes d < LS_PT)
            if (a.head_w && d < LS_PT && nh < N) {
                t_lm = a.lm_fp[nh]; t_lp0 = a.last_pos[2 * nh]; t_lp1 = a.last_pos[2 * nh + 1];
                if (a.tt) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) t_acc[k] = a.acc[(size_t)nh * 5 + k];
                }
            }
        }
        int rbase[LS_PT]; // row of slice 0 of this thread's nodes (slice t is rbase + t * H); out-of-range nodes clamp to the last one (never stored)
#pragma unroll
        for (int k = 0; k < LS_PT; ++k) {
            const int n = min(n0 + q + 8 * k, N - 1);
            const int e = n / H;
            rbase[k] = e * S * H + (n - e * H);
        }
#pragma unroll
        for (int k = 0; k < LS_PT; ++k) { // all loads issued back to back, no control flow between them
            const int n = min(n0 + q + 8 * k, N - 1);
            creg[k] = a.zero_state ? 0.0f : a.c[(size_t)n * 64 + d];
            hreg[k] = a.zero_state ? 0.0f : a.h[(size_t)n * 64 + d];
            xn[k] = a.in_mask[rbase[k]] * a.xs[(size_t)rbase[k] * 64 + d]; // m * x_0 ((x m) W = m (x W) for m in {0, 1})
        }
#pragma unroll
        for (int k = 0; k < LS_PT; ++k) X[(q + 8 * k) * LS_SX + 64 + d] = hreg[k];
        GL_T(8);
        for (int t = 0; t < S; ++t) {
#pragma unroll
            for (int k = 0; k < LS_PT; ++k) X[(q + 8 * k) * LS_SX + d] = xn[k];
            __syncthreads();
            GL_T(9);
            if (t + 1 < S) { // the next slice's rows travel while this slice is computed
#pragma unroll
                for (int k = 0; k < LS_PT; ++k) {
                    const int r = rbase[k] + (t + 1) * H;
                    xn[k] = a.in_mask[r] * a.xs[(size_t)r * 64 + d];
                }
            }
            // gates[node][f] for the feature blocks wave8 and wave8 + 8, all four row tiles
            f32x4 bv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const f32x4 *>(B + (wave8 + 8 * j) * 16 + 4 * g);
#pragma unroll
            for (int rt = 0; rt < LS_ROWS / 16; ++rt) {
                f32x4 acc[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float *inr = X + (16 * rt + i) * LS_SX + 8 * g;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4 *>(inr + 32 * ks), x1 = *reinterpret_cast<const f32x4 *>(inr + 32 * ks + 4);
                    bf16x8 bh, bl;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const __bf16 h0 = (__bf16)x0[u], h1 = (__bf16)x1[u];
                        bh[u] = h0; bh[4 + u] = h1;
                        bl[u] = (__bf16)(x0[u] - (float)h0); bl[4 + u] = (__bf16)(x1[u] - (float)h1);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.l[ks][j], bh, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[ks][j], bl, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h[ks][j], bh, acc[j], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<f32x4 *>(G + (16 * rt + i) * LS_SG + (wave8 + 8 * j) * 16 + 4 * g) = acc[j] + bv[j];
            }
            __syncthreads();
            GL_T(10);
            // the cell (PyTorch gate order i, f, g, o), decode-step blend h = h' m + h (1 - m), post mask after the last slice
            const bool last = t == S - 1;
#pragma unroll
            for (int k = 0; k < LS_PT; ++k) {
                const int nl = q + 8 * k, n = n0 + nl;
                const float *gr = G + nl * LS_SG;
                const float gi = gr[d], gf = gr[64 + d], gg = gr[128 + d], go = gr[192 + d];
                const float c0 = creg[k], h0 = X[nl * LS_SX + 64 + d];
                float cn = gl_sigmoid(gf) * c0 + gl_sigmoid(gi) * gl_tanh(gg);
                float hn = gl_sigmoid(go) * gl_tanh(cn);
                if (a.blend_mask && n < N) { const float bm = a.blend_mask[n]; hn = hn * bm + h0 * (1.0f - bm); cn = cn * bm + c0 * (1.0f - bm); }
                if (last && a.post_mask && n < N) { const float pm = a.post_mask[n]; hn *= pm; cn *= pm; }
                creg[k] = cn;
                X[nl * LS_SX + 64 + d] = hn;
            }
            // (the next slice's staging only writes the x half of X, which nobody reads before the barrier that follows it)
            GL_T(11);
        }
#pragma unroll
        for (int k = 0; k < LS_PT; ++k) {
            const int nl = q + 8 * k, n = n0 + nl;
            if (n < N) { a.c[(size_t)n * 64 + d] = creg[k]; a.h[(size_t)n * 64 + d] = X[nl * LS_SX + 64 + d]; }
        }
        if (a.head_w) {
            // the head on the hidden state this thread's wavefront just wrote (node rows q, q + 8, ...: lane = hidden unit)
            float raw[5] = {0.f, 0.f, 0.f, 0.f, 0.f}; // lane k keeps node row q + 8 k, so the eight scalar tails below run side by side
#pragma unroll
            for (int k = 0; k < LS_PT; ++k) {
                const float hv = X[(q + 8 * k) * LS_SX + 64 + d];
                const float s0 = wv_sum(hv * hw_[0]), s1 = wv_sum(hv * hw_[1]), s2 = wv_sum(hv * hw_[2]), s3 = wv_sum(hv * hw_[3]), s4 = wv_sum(hv * hw_[4]);
                if (d == k) { raw[0] = s0; raw[1] = s1; raw[2] = s2; raw[3] = s3; raw[4] = s4; }
            }
            const int n = n0 + q + 8 * d;
            if (d < LS_PT && n < N) {
#pragma unroll
                for (int k = 0; k < 5; ++k) raw[k] += hb_[k];
                const int tt = a.tt;
                const float lm = t_lm;
                const float sx = expf(raw[2]), sy = expf(raw[3]), corr = tanhf(raw[4]);
                float *ac = a.acc + (size_t)n * 5;
                const float a0 = (tt ? t_acc[0] : 0.f) + raw[0], a1 = (tt ? t_acc[1] : 0.f) + raw[1];
                const float a2 = (tt ? t_acc[2] : 0.f) + sx * sx, a3 = (tt ? t_acc[3] : 0.f) + sy * sy, a4 = (tt ? t_acc[4] : 0.f) + corr * sx * sy;
                ac[0] = a0; ac[1] = a1; ac[2] = a2; ac[3] = a3; ac[4] = a4;
                const float sxc = sqrtf(a2), syc = sqrtf(a3);
                float *o = a.out_traj + ((size_t)n * GP + tt) * 5;
                o[0] = (a0 + t_lp0) * lm + GST_INVALID * (1.0f - lm);
                o[1] = (a1 + t_lp1) * lm + GST_INVALID * (1.0f - lm);
                o[2] = sxc; o[3] = syc; o[4] = a4 / (sxc * syc);
                a.x_sample[2 * n] = raw[0] * lm; a.x_sample[2 * n + 1] = raw[1] * lm; // sampling = False: the mean, masked for the next step
            }
        }
        __syncthreads(); // the next tile refills X
        GL_T(12);
    }
}

// ---- VecPretextNormalize ----
// history push: human_pos = robot_xy + spatial_edges[:, :, :2] (fp32), visibility mask, into ring slot `slot`
__global__ __launch_bounds__(256) void pretext_push_kernel(int E, int H, int D, const float *__restrict__ robot_node, const float *__restrict__ se,
                                                           const uint8_t *__restrict__ vis, float *__restrict__ ring_traj, uint8_t *__restrict__ ring_mask,
                                                           int slot)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * H) return;
    const int e = idx / H;
    float *t = ring_traj + ((size_t)slot * E * H + idx) * 2;
    t[0] = robot_node[e * 7] + se[(size_t)idx * D];
    t[1] = robot_node[e * 7 + 1] + se[(size_t)idx * D + 1];
    ring_mask[(size_t)slot * E * H + idx] = vis[idx] ? 1 : 0;
}
__global__ void pretext_fill_kernel(size_t n, float *__restrict__ ring_traj, uint8_t *__restrict__ ring_mask)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { ring_traj[2 * i] = GST_INVALID; ring_traj[2 * i + 1] = GST_INVALID; ring_mask[i] = 0; }
}
// social penalty, prediction write-back into spatial_edges[:, :, 2:], stable sort by current distance: one wavefront per env
__global__ __launch_bounds__(64) void pretext_post_kernel(int E, int H, int D, const float *__restrict__ robot_node, const float *__restrict__ se_in,
                                                          const float *__restrict__ out_traj, const float *__restrict__ out_mask, float dist,
                                                          float collision_penalty, float *__restrict__ rews, float *__restrict__ se_out)
{
    const int e = blockIdx.x, lane = threadIdx.x;
    const bool isH = lane < H;
    const int hl = isH ? lane : 0;
    const float rx = robot_node[e * 7], ry = robot_node[e * 7 + 1];
    const float *si = se_in + ((size_t)e * H + hl) * D;
    const float om = out_mask[(size_t)e * H + hl];
    float row[2 * (GP + 1)];
    row[0] = si[0]; row[1] = si[1];
    float pen = 0.0f;
#pragma unroll
    for (int k = 0; k < GP; ++k) {
        const float *o = out_traj + (((size_t)e * H + hl) * GP + k) * 5;
        const float dx = o[0] - rx, dy = o[1] - ry;
        const bool coll = isH && om != 0.0f && sqrtf(dx * dx + dy * dy) < dist;
        const float pk = collision_penalty / (float)(1 << (k + 2));
        if (coll && pk < pen) pen = pk;
        row[2 + 2 * k] = om != 0.0f ? dx : si[2 + 2 * k];
        row[3 + 2 * k] = om != 0.0f ? dy : si[3 + 2 * k];
    }
    const float pmin = wv_min(pen);
    if (lane == 0) rews[e] += pmin;
    const float key = isH ? sqrtf(row[0] * row[0] + row[1] * row[1]) : INFINITY;
    int rank = 0;
    for (int m = 0; m < H; ++m) {
        const float km = wv_readlane(key, m);
        rank += (km < key || (km == key && m < lane)) ? 1 : 0;
    }
    if (isH) {
        float *so = se_out + ((size_t)e * H + rank) * D;
#pragma unroll
        for (int d = 0; d < 2 * (GP + 1); ++d) so[d] = row[d];
    }
}

// The observation window of one call is the previous call's window moved on by one frame (the wrapper's history ring), and the spatial encoding
// of a frame -- NodeEncoderLayer over the H nodes of one (env, frame) group -- is a function of that group's inputs only: displacements
// (x_t - x_{t-1}) m and masks m = mask[t-1] mask[T-1].  Frame t of this call (t = 1 .. T-2) therefore has the encoding frame t + 1 had in the
// previous call whenever its H inputs and masks are the same 3 H numbers -- which they are unless a human's visibility in the NEWEST frame
// changed (that mask multiplies every frame's) or the env was reset.  One wavefront per group compares them bit for bit: equal -> the 64 H
// encoded values are copied over from the previous call's buffer, anything else (and always frame 0, whose displacement is defined as 0, and the
// newest frame) -> the group goes on the list gst_layer_kernel works through.  Same numbers either way: a row's encoding does not depend on
// which other groups share its tile.
// The list: frames 0 and T - 1 of every env at fixed places (entries [0, 2 E)), the other groups that could not be taken over behind them -- one
// device-scope atomic per workgroup of 16 groups (same-address atomics serialise in the L2 at ~12 ns each: one per group took 60 us).
__global__ __launch_bounds__(1024) void gst_reuse_kernel(int E, int H, int have_prev, const float *__restrict__ rel, const float *__restrict__ m_rel,
                                                         const float *__restrict__ rel_prev, const float *__restrict__ m_prev, const float *__restrict__ xs_prev,
                                                         float *__restrict__ xs, int *__restrict__ glist, int *__restrict__ gcount)
{
    __shared__ int s_cnt, s_base;
    const int lane = threadIdx.x & 63;
    const int gidx = blockIdx.x * 16 + (threadIdx.x >> 6); // group = (env e, frame t): rows gidx * H .. + H; one wavefront each
    const bool live = gidx < E * GT;
    const int e = gidx / GT, t = gidx - e * GT;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    bool same = live && have_prev && t >= 1 && t <= GT - 2;
    if (same) {
        const size_t r = (size_t)gidx * H + lane, rp = r + H; // the same env's frame t + 1 of the previous call
        bool eq = true;
        if (lane < H) {
            eq = __float_as_uint(rel[2 * r]) == __float_as_uint(rel_prev[2 * rp]) && __float_as_uint(rel[2 * r + 1]) == __float_as_uint(rel_prev[2 * rp + 1]) &&
                 __float_as_uint(m_rel[r]) == __float_as_uint(m_prev[rp]);
        }
        same = __ballot(!eq) == 0ull;
    }
    const bool listed = live && !same && t >= 1 && t <= GT - 2;
    int slot = 0;
    if (listed && lane == 0) slot = atomicAdd(&s_cnt, 1);
    if (same) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(xs_prev + ((size_t)gidx + 1) * H * 64);
        f32x4 *dst = reinterpret_cast<f32x4 *>(xs + (size_t)gidx * H * 64);
        const int n = H * 16;
        for (int k0 = lane; k0 < n; k0 += 64 * 5) { // (H = 20: the group's 320 float4 in one go, all loads in flight before the first store)
            f32x4 v[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) if (k0 + 64 * u < n) v[u] = __builtin_nontemporal_load(src + k0 + 64 * u);
#pragma unroll
            for (int u = 0; u < 5; ++u) if (k0 + 64 * u < n) dst[k0 + 64 * u] = v[u];
        }
    }
    if (live && lane == 0) {
        if (t == 0) glist[e] = gidx;
        if (t == GT - 1) glist[E + e] = gidx;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt > 0) s_base = atomicAdd(gcount, s_cnt);
    __syncthreads();
    if (listed && lane == 0) glist[2 * E + s_base + slot] = gidx;
}

constexpr size_t g_align(size_t x) { return (x + 255) & ~size_t(255); }

} // namespace

struct cn_gst {
    int H, maxE;
    bool weights_set;
    char *blob;
    float *emb_w, *emb_b, *in_w, *in_b, *out_w, *out_b, *n_w, *n_b, *n1_w, *n1_b, *l1_w, *l1_b, *l2_w, *l2_b;
    float *wih, *whh, *bih, *bhh, *h2p_w, *h2p_b;
    float *w_cat, *f_lstm; // [256,128] = [W_ih | W_hh] and its fragment image (gst_lstm_kernel)
    float *f_in, *f_out, *f_l1, *f_l2; // bf16 hi/lo MFMA-fragment images of in_proj / out_proj / linear1 / linear2 (gst_layer_kernel)
    // workspace (rows = maxE * 5 * H)
    float *m_rel, *lm_fp, *rel, *last_pos, *xs, *h, *c, *acc, *x_sample;
    float *out_traj, *out_mask; // internal buffers for the wrapper path
    // the previous call's observation-period inputs and encodings (gst_reuse_kernel); xs_dec: the decode steps' encoded rows (they must not
    // overwrite the window's)
    float *m_rel_prev, *rel_prev, *xs_prev, *xs_dec;
    int *glist, *gcount;
    int prev_E;        // envs of the call whose window the *_prev buffers hold (0 = none: first call, new weights)
    bool reuse;        // CN_GST_REUSE=0 switches the reuse off (A/B, tests)
    // VecPretextNormalize history
    // (its own allocation: the length depends on the prediction stride, cn_gst_wrapper_set_interval)
    float *ring_traj;   // [ring_len][maxE][H][2]
    uint8_t *ring_mask; // [ring_len][maxE][H]
    int ring_E, ring_pos;
    int interval, ring_len; // pred_interval = int(pred_timestep // time_step) and buffer_len = (GT - 1) * interval + 1 (vec_pretext_normalize.py:56-57)
};

static int gst_alloc_ring(cn_gst *g, int interval)
{
    const int len = (GT - 1) * interval + 1;
    const size_t n = (size_t)len * g->maxE * g->H;
    float *t = nullptr;
    uint8_t *m = nullptr;
    CN_HIP(hipMalloc((void **)&t, n * 2 * sizeof(float)));
    if (hipMalloc((void **)&m, n) != hipSuccess) { (void)hipFree(t); cn_set_error("cn_gst: hipMalloc of the history ring failed"); return CN_ERR_HIP; }
    if (g->ring_traj) (void)hipFree(g->ring_traj);
    if (g->ring_mask) (void)hipFree(g->ring_mask);
    g->ring_traj = t; g->ring_mask = m; g->interval = interval; g->ring_len = len; g->ring_E = 0; g->ring_pos = 0;
    return CN_OK;
}

extern "C" int cn_gst_create(int human_num, int max_envs, cn_gst **out)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(out && human_num >= 1 && human_num <= CN_MAX_HUMANS && max_envs >= 1, "cn_gst_create: bad argument");
    cn_gst *g = new (std::nothrow) cn_gst{};
    CN_REQUIRE(g, "cn_gst_create: out of host memory");
    g->H = human_num; g->maxE = max_envs;
    const size_t N = (size_t)max_envs * human_num, R = N * GT;
    size_t off = 0;
    auto carve = [&](size_t nfloat) { size_t o = off; off += g_align(nfloat * sizeof(float)); return o; };
    const size_t o_w[] = {carve(128), carve(64), carve(192 * 64), carve(192), carve(64 * 64), carve(64), carve(64), carve(64), carve(64), carve(64),
                          carve(128 * 64), carve(128), carve(64 * 128), carve(64), carve(256 * 64), carve(256 * 64), carve(256), carve(256), carve(320), carve(5)};
    const size_t o_mrel = carve(R), o_lm = carve(N), o_rel = carve(2 * R), o_lp = carve(2 * N), o_xs = carve(R * 64);
    const size_t o_h = carve(N * 64), o_c = carve(N * 64), o_acc = carve(N * 5), o_xsamp = carve(N * 2), o_ot = carve(N * GP * 5), o_om = carve(N);
    const size_t o_wcat = carve(256 * 128), o_flstm = carve(256 * 128);
    const size_t o_mrelp = carve(R), o_relp = carve(2 * R), o_xsp = carve(R * 64), o_xsd = carve(N * 64), o_gl = carve(R / human_num + 64);
    const size_t o_fin = carve(192 * 64), o_fout = carve(64 * 64), o_fl1 = carve(128 * 64), o_fl2 = carve(64 * 128);
    char *base = nullptr;
    hipError_t herr = hipMalloc((void **)&base, off);
    if (herr != hipSuccess) { delete g; cn_set_error("cn_gst_create: hipMalloc(%zu) failed: %s", off, hipGetErrorString(herr)); return CN_ERR_HIP; }
    g->blob = base;
    auto F = [&](size_t o) { return (float *)(base + o); };
    float **wp[] = {&g->emb_w, &g->emb_b, &g->in_w, &g->in_b, &g->out_w, &g->out_b, &g->n_w, &g->n_b, &g->n1_w, &g->n1_b,
                    &g->l1_w, &g->l1_b, &g->l2_w, &g->l2_b, &g->wih, &g->whh, &g->bih, &g->bhh, &g->h2p_w, &g->h2p_b};
    for (int i = 0; i < 20; ++i) *wp[i] = F(o_w[i]);
    g->m_rel = F(o_mrel); g->lm_fp = F(o_lm); g->rel = F(o_rel); g->last_pos = F(o_lp); g->xs = F(o_xs); g->h = F(o_h); g->c = F(o_c);
    g->acc = F(o_acc); g->x_sample = F(o_xsamp); g->out_traj = F(o_ot); g->out_mask = F(o_om);
    g->w_cat = F(o_wcat); g->f_lstm = F(o_flstm);
    g->f_in = F(o_fin); g->f_out = F(o_fout); g->f_l1 = F(o_fl1); g->f_l2 = F(o_fl2);
    g->m_rel_prev = F(o_mrelp); g->rel_prev = F(o_relp); g->xs_prev = F(o_xsp); g->xs_dec = F(o_xsd);
    g->glist = reinterpret_cast<int *>(F(o_gl)) + 64; g->gcount = reinterpret_cast<int *>(F(o_gl));
    g->prev_E = 0;
    g->reuse = !(getenv("CN_GST_REUSE") && atoi(getenv("CN_GST_REUSE")) == 0);
    g->weights_set = false;
    if (int rc = gst_alloc_ring(g, 1)) { (void)hipFree(base); delete g; return rc; }
    *out = g;
    return CN_OK;
}

extern "C" int cn_gst_destroy(cn_gst *g)
{
    if (!g) return CN_OK;
    if (g->ring_traj) (void)hipFree(g->ring_traj);
    if (g->ring_mask) (void)hipFree(g->ring_mask);
    if (g->blob) CN_HIP(hipFree(g->blob));
    delete g;
    return CN_OK;
}

extern "C" int cn_gst_set_weights(cn_gst *g, const cn_gst_weights *w, void *stream)
{
    CN_REQUIRE(g && w, "cn_gst_set_weights: null argument");
    const float *const *src = reinterpret_cast<const float *const *>(w);
    const size_t n[20] = {128, 64, 192 * 64, 192, 64 * 64, 64, 64, 64, 64, 64, 128 * 64, 128, 64 * 128, 64, 256 * 64, 256 * 64, 256, 256, 320, 5};
    float *dst[20] = {g->emb_w, g->emb_b, g->in_w, g->in_b, g->out_w, g->out_b, g->n_w, g->n_b, g->n1_w, g->n1_b,
                      g->l1_w, g->l1_b, g->l2_w, g->l2_b, g->wih, g->whh, g->bih, g->bhh, g->h2p_w, g->h2p_b};
    for (int i = 0; i < 20; ++i) {
        CN_REQUIRE(src[i] != nullptr, "cn_gst_set_weights: weight pointer #%d is null", i);
        CN_HIP(hipMemcpyAsync(dst[i], src[i], n[i] * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = rn_fused_bake(192, 64, g->in_w, g->f_in, st)) || (rc = rn_fused_bake(64, 64, g->out_w, g->f_out, st)) ||
        (rc = rn_fused_bake(128, 64, g->l1_w, g->f_l1, st)) || (rc = rn_fused_bake(64, 128, g->l2_w, g->f_l2, st))) return rc;
    CN_HIP(hipMemcpy2DAsync(g->w_cat, 128 * sizeof(float), g->wih, 64 * sizeof(float), 64 * sizeof(float), 256, hipMemcpyDeviceToDevice, st));
    CN_HIP(hipMemcpy2DAsync(g->w_cat + 64, 128 * sizeof(float), g->whh, 64 * sizeof(float), 64 * sizeof(float), 256, hipMemcpyDeviceToDevice, st));
    if ((rc = rn_fused_bake(256, 128, g->w_cat, g->f_lstm, st))) return rc;
    g->weights_set = true;
    g->prev_E = 0; // encodings of the old weights are not this model's
    return CN_OK;
}

#ifdef GST_TIMING
extern "C" int cn_gst_set_timing(long long *buf)
{
    CN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gst_tim), &buf, sizeof(buf)));
    return CN_OK;
}
#endif

// NodeEncoderLayer over `rows` rows in groups of H nodes: x2 [rows,2], mask [rows] -> g->xs [rows,64] (one launch)
static int gst_layer(cn_gst *g, int rows, const float *x2, const float *mask, hipStream_t st, float *xs_out = nullptr, bool listed = false)
{
    const int H = g->H;
    int TG = GL_ROWS / H; TG = TG < 1 ? 1 : TG;
    CN_REQUIRE(TG * H <= GL_ROWS, "cn_gst: more than %d humans per group is not supported by the fused encoder layer", GL_ROWS);
    static thread_local int attr_dev = -1;
    int dev = 0;
    CN_HIP(hipGetDevice(&dev));
    if (dev != attr_dev) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gst_layer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GL_LDS_FLOATS * sizeof(float))));
        attr_dev = dev;
    }
    const int n_tiles = (rows + TG * H - 1) / (TG * H);
    GstLayerArgs a{x2, mask, g->emb_w, g->emb_b, g->n_w, g->n_b, g->n1_w, g->n1_b, g->f_in, g->in_b, g->f_out, g->out_b, g->f_l1, g->l1_b, g->f_l2, g->l2_b,
                   xs_out ? xs_out : g->xs, listed ? g->glist : nullptr, listed ? g->gcount : nullptr, listed ? 2 * (rows / (GT * H)) : 0};
    hipLaunchKernelGGL(gst_layer_kernel, dim3(n_tiles < 256 * GL_WGS ? n_tiles : 256 * GL_WGS), dim3(GL_THREADS), GL_LDS_FLOATS * sizeof(float), st, rows, H, TG, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// the LSTM over S slices of the encoded rows g->xs (S = 5: the observation period from h = c = 0 set by the caller; S = 1: one decode step)
static int gst_lstm(cn_gst *g, int E, int S, const float *in_mask, const float *blend_mask, const float *post_mask, int tt, float *out_traj, hipStream_t st,
                    int zero_state = 0, const float *xs_in = nullptr)
{
    static thread_local int attr_dev = -1;
    int dev = 0;
    CN_HIP(hipGetDevice(&dev));
    if (dev != attr_dev) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gst_lstm_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ls_lds_floats(32) * sizeof(float))));
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gst_lstm_kernel<80>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ls_lds_floats(80) * sizeof(float))));
        attr_dev = dev;
    }
    GstLstmArgs a{xs_in ? xs_in : g->xs, in_mask, g->f_lstm, g->bih, g->bhh, g->h, g->c, blend_mask, post_mask, tt, zero_state, g->h2p_w, g->h2p_b, g->lm_fp, g->last_pos,
                  g->acc, out_traj, g->x_sample};
    // Tile size: a pass over an 80-node tile costs ~2.25x a pass over a 32-node one (measured at 2048 envs x 20 nodes: 2 tiles of 80 per workgroup
    // 0.716 ms per step against 5 tiles of 32 0.745, 64-node tiles 0.767: 640 tiles on 256 workgroups leave a third of them idle in the last
    // turn), so the larger tile wins where it does not cost a whole extra turn of the workgroups.  CN_LS_ROWS=32 / 80 forces one (A/B).
    static const int forced = getenv("CN_LS_ROWS") ? atoi(getenv("CN_LS_ROWS")) : CN_LS_ROWS;
    const int N = E * g->H, wgs = 256 * CN_GST_WGS;
    const int t32 = (N + 31) / 32, t80 = (N + 79) / 80;
    const double c32 = (double)((t32 + wgs - 1) / wgs), c80 = 2.25 * (double)((t80 + wgs - 1) / wgs);
    const bool big = forced == 80 || (forced != 32 && c80 < c32);
    if (big) hipLaunchKernelGGL(gst_lstm_kernel<80>, dim3(t80 < wgs ? t80 : wgs), dim3(512), ls_lds_floats(80) * sizeof(float), st, E, g->H, S, a);
    else hipLaunchKernelGGL(gst_lstm_kernel<32>, dim3(t32 < wgs ? t32 : wgs), dim3(512), ls_lds_floats(32) * sizeof(float), st, E, g->H, S, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

static int gst_forward(cn_gst *g, int E, const float *traj, long long se, long long sh, long long stt, const uint8_t *mask_u8, const float *mask_f,
                       long long me, long long mh, long long mt, int rot, int step, int ring, float *out_traj, float *out_mask, hipStream_t st)
{
    if (!g->weights_set) { cn_set_error("cn_gst: call cn_gst_set_weights first"); return CN_ERR_STATE; }
    const int H = g->H, N = E * H, R = N * GT;
    int rc;
    // the previous call's window becomes `prev`, this call's is written over the one before it
    if (g->reuse) { std::swap(g->rel, g->rel_prev); std::swap(g->m_rel, g->m_rel_prev); std::swap(g->xs, g->xs_prev); }
    hipLaunchKernelGGL(gst_obs_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, st, E, H, traj, se, sh, stt, mask_u8, mask_f, me, mh, mt, rot, step, ring,
                       g->m_rel, g->lm_fp, g->rel, g->last_pos);
    CN_CHECK_LAUNCH();
    // observation period: spatial encoding of all 5 slices at once, then the LSTM over time
    if (g->reuse) {
        CN_HIP(hipMemsetAsync(g->gcount, 0, sizeof(int), st));
        hipLaunchKernelGGL(gst_reuse_kernel, dim3((E * GT + 15) / 16), dim3(1024), 0, st, E, H, g->prev_E == E ? 1 : 0, g->rel, g->m_rel, g->rel_prev, g->m_rel_prev, g->xs_prev,
                           g->xs, g->glist, g->gcount);
        CN_CHECK_LAUNCH();
        if ((rc = gst_layer(g, R, g->rel, g->m_rel, st, g->xs, true))) return rc;
        g->prev_E = E;
    } else if ((rc = gst_layer(g, R, g->rel, g->m_rel, st))) return rc;
    if ((rc = gst_lstm(g, E, GT, g->m_rel, nullptr, g->lm_fp, 0, out_traj, st, 1))) return rc; // from h = c = 0; + the head of decode step 0
    // prediction period (recursive decoding on the mean)
    for (int tt = 1; tt < GP; ++tt) {
        if ((rc = gst_layer(g, N, g->x_sample, g->lm_fp, st, g->xs_dec))) return rc;
        if ((rc = gst_lstm(g, E, 1, g->lm_fp, g->lm_fp, nullptr, tt, out_traj, st, 0, g->xs_dec))) return rc; // + the head of decode step tt
    }
    if (out_mask != g->lm_fp) CN_HIP(hipMemcpyAsync(out_mask, g->lm_fp, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, st));
    return CN_OK;
}

extern "C" int cn_gst_predict(cn_gst *g, int E, const float *in_traj, const float *in_mask, float *out_traj, float *out_mask, void *stream)
{
    CN_REQUIRE(g && in_traj && in_mask && out_traj && out_mask && E >= 1 && E <= g->maxE, "cn_gst_predict: bad argument");
    const long long H = g->H;
    return gst_forward(g, E, in_traj, H * GT * 2, GT * 2, 2, nullptr, in_mask, H * GT, GT, 1, 0, 1, GT, out_traj, out_mask, (hipStream_t)stream);
}

extern "C" int cn_gst_wrapper_set_interval(cn_gst *g, int pred_interval)
{
    CN_REQUIRE(g && pred_interval >= 1 && pred_interval <= 64, "cn_gst_wrapper_set_interval: pred_interval must be in [1,64]");
    if (pred_interval == g->interval) return CN_OK;
    CN_HIP(hipDeviceSynchronize()); // the old ring may still be read by work in flight (a configuration call, never on the hot path)
    return gst_alloc_ring(g, pred_interval);
}

extern "C" int cn_gst_wrapper_history_len(const cn_gst *g) { return g ? g->ring_len : 0; }

extern "C" int cn_gst_wrapper_reset(cn_gst *g, int E, void *stream)
{
    CN_REQUIRE(g && E >= 1 && E <= g->maxE, "cn_gst_wrapper_reset: bad argument");
    const size_t n = (size_t)g->ring_len * E * g->H;
    hipLaunchKernelGGL(pretext_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, g->ring_traj, g->ring_mask);
    CN_CHECK_LAUNCH();
    g->ring_E = E; g->ring_pos = 0;
    return CN_OK;
}

// history in TIME order (oldest observation first), whatever the ring's rotation: traj [ring_len,E,H,2] fp32, mask [ring_len,E,H] u8
extern "C" int cn_gst_wrapper_save(cn_gst *g, float *traj, uint8_t *mask, void *stream)
{
    CN_REQUIRE(g && traj && mask, "cn_gst_wrapper_save: null argument");
    if (g->ring_E < 1) { cn_set_error("cn_gst_wrapper_save: no history (call cn_gst_wrapper_reset first)"); return CN_ERR_STATE; }
    const size_t N = (size_t)g->ring_E * g->H;
    for (int t = 0; t < g->ring_len; ++t) {
        const int slot = (g->ring_pos + t) % g->ring_len; // ring_pos = the slot the next push overwrites = the oldest
        CN_HIP(hipMemcpyAsync(traj + (size_t)t * N * 2, g->ring_traj + (size_t)slot * N * 2, N * 2 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        CN_HIP(hipMemcpyAsync(mask + (size_t)t * N, g->ring_mask + (size_t)slot * N, N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return CN_OK;
}

extern "C" int cn_gst_wrapper_load(cn_gst *g, int E, const float *traj, const uint8_t *mask, void *stream)
{
    CN_REQUIRE(g && traj && mask && E >= 1 && E <= g->maxE, "cn_gst_wrapper_load: bad argument");
    const size_t n = (size_t)g->ring_len * E * g->H;
    CN_HIP(hipMemcpyAsync(g->ring_traj, traj, n * 2 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    CN_HIP(hipMemcpyAsync(g->ring_mask, mask, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    g->ring_E = E; g->ring_pos = 0; // slot t = time t: the next push overwrites slot 0, the oldest
    return CN_OK;
}

extern "C" int cn_gst_wrapper_step(cn_gst *g, int E, const cn_obs *obs, float robot_plus_human_radius, float collision_penalty, float *rewards,
                                   float *spatial_edges_out, void *stream)
{
    CN_REQUIRE(g && obs && obs->robot_node && obs->spatial_edges && obs->visible_masks && spatial_edges_out, "cn_gst_wrapper_step: null argument");
    if (g->ring_E != E) { cn_set_error("cn_gst_wrapper_step: call cn_gst_wrapper_reset(E=%d) first", E); return CN_ERR_STATE; }
    hipStream_t st = (hipStream_t)stream;
    const int H = g->H, D = 2 * (GP + 1), N = E * H;
    // deque.append: the oldest slot is overwritten; time order = slots (pos+1 .. pos+len) % len, of which every interval-th is read
    // (in_traj[:, :, ::pred_interval]: the oldest, ..., the newest -- len = (GT-1) * interval + 1)
    hipLaunchKernelGGL(pretext_push_kernel, dim3((N + 255) / 256), dim3(256), 0, st, E, H, D, obs->robot_node, obs->spatial_edges, obs->visible_masks,
                       g->ring_traj, g->ring_mask, g->ring_pos);
    CN_CHECK_LAUNCH();
    const int rot = (g->ring_pos + 1) % g->ring_len;
    g->ring_pos = rot;
    if (int rc = gst_forward(g, E, g->ring_traj, (long long)H * 2, 2, (long long)N * 2, g->ring_mask, nullptr, H, 1, N, rot, g->interval, g->ring_len,
                             g->out_traj, g->lm_fp, st)) return rc;
    float *rw = rewards;
    hipLaunchKernelGGL(pretext_post_kernel, dim3(E), dim3(64), 0, st, E, H, D, obs->robot_node, obs->spatial_edges, g->out_traj, g->lm_fp,
                       robot_plus_human_radius, collision_penalty, rw ? rw : g->acc /*scratch*/, spatial_edges_out);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
