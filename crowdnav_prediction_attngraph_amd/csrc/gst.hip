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
// row = (env * S + t) * H + human.  Dense contractions (K = 64 / 128) run on the fp32 MFMA GEMM of gemm.h; everything
// between them is wave-per-row / wave-per-(env, t) kernels.  All work is enqueued on the caller's stream.
#include "common.h"
#include "gemm.h"

#include <cmath>
#include <new>

namespace {

constexpr int GT = 5, GP = 5; // obs_seq_len, pred_seq_len
constexpr float GST_INVALID = -999.0f;

// masks + displacements (crowd_nav_interface_parallel.py:71-89).  traj element (e,h,t) lives at
// traj[e*se + h*sh + ((t + rot) % GT)*st] (+0/+1 for x/y); mask likewise with strides me/mh/mt.
// NOTE loss_mask_rel_obs[t>=1] = mask[t-1] * mask[T-1] exactly as written in the reference (:76).
__global__ __launch_bounds__(256) void gst_obs_prep_kernel(int E, int H, const float *__restrict__ traj, long long se, long long sh, long long st,
                                                           const uint8_t *__restrict__ mask_u8, const float *__restrict__ mask_f,
                                                           long long me, long long mh, long long mt, int rot, float *__restrict__ m_rel,
                                                           float *__restrict__ lm_fp, float *__restrict__ rel, float *__restrict__ last_pos)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * H) return;
    const int e = idx / H, h = idx - e * H;
    float m[GT], px[GT], py[GT];
#pragma unroll
    for (int t = 0; t < GT; ++t) {
        const int tt = (t + rot) % GT;
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

// node_embedding (2 -> 64) + LayerNorm(norm_node) + pedestrian mask: one wavefront per row, lane = feature.
// attn_mask_ped = (row of the adjacency has any 1) == the row's own mask (adjacency = outer product of the masks).
__global__ __launch_bounds__(256) void gst_embed_ln_kernel(int rows, const float *__restrict__ x2, const float *__restrict__ mask,
                                                           const float *__restrict__ We, const float *__restrict__ be,
                                                           const float *__restrict__ g, const float *__restrict__ b, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float v = We[2 * lane] * x2[2 * r] + We[2 * lane + 1] * x2[2 * r + 1] + be[lane];
    const float mean = wv_sum(v) * (1.0f / 64.0f);
    const float d = v - mean;
    const float var = wv_sum(d * d) * (1.0f / 64.0f);
    out[(size_t)r * 64 + lane] = (d * rsqrtf(var + 1e-5f) * g[lane] + b[lane]) * mask[r];
}

// plain LayerNorm(norm1_node), one wavefront per row
__global__ __launch_bounds__(256) void gst_ln_kernel(int rows, const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ b,
                                                     float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float v = x[(size_t)r * 64 + lane];
    const float mean = wv_sum(v) * (1.0f / 64.0f);
    const float d = v - mean;
    const float var = wv_sum(d * d) * (1.0f / 64.0f);
    out[(size_t)r * 64 + lane] = d * rsqrtf(var + 1e-5f) * g[lane] + b[lane];
}

// VanillaMultiheadAttention core for one group (env, time slice) of H nodes: 8 heads x 8 dims.
// p = softmax(q k^T / sqrt(8)) over ALL nodes, then p *= m_i * m_j, p /= (sum_j p + 1e-10)   (mha.py:236-242).
// One wavefront per group; K and V rows in LDS; lanes enumerate (node i, head) pairs.
__global__ __launch_bounds__(256) void gst_attention_kernel(int groups, int H, const float *__restrict__ qkv, const float *__restrict__ mask,
                                                            float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * (blockDim.x >> 6) + wave;
    if (g >= groups) return;
    float *Ks = smem + (size_t)wave * (2 * H * 65 + 64);
    float *Vs = Ks + H * 65;
    float *Ms = Vs + H * 65;
    const float *base = qkv + (size_t)g * H * 192;
    for (int j = 0; j < H; ++j) {
        Ks[j * 65 + lane] = base[(size_t)j * 192 + 64 + lane];
        Vs[j * 65 + lane] = base[(size_t)j * 192 + 128 + lane];
    }
    if (lane < H) Ms[lane] = mask[(size_t)g * H + lane];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const float scale = 0.35355339059327373f; // 8^-0.5
    for (int pq = lane; pq < H * 8; pq += 64) {
        const int i = pq >> 3, hd = pq & 7;
        float q[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) q[d] = base[(size_t)i * 192 + hd * 8 + d] * scale;
        float mx = -INFINITY;
        for (int j = 0; j < H; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) s += q[d] * Ks[j * 65 + hd * 8 + d];
            mx = fmaxf(mx, s);
        }
        float Z = 0.0f, Zm = 0.0f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < H; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) s += q[d] * Ks[j * 65 + hd * 8 + d];
            const float ex = expf(s - mx);
            Z += ex;
            const float em = ex * Ms[j];
            Zm += em;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] += em * Vs[j * 65 + hd * 8 + d];
        }
        // softmax p_j = ex_j / Z ; masked p'_j = p_j m_i m_j ; renormalised by (sum_j p'_j + 1e-10)
        const float mi = Ms[i];
        const float denom = mi * Zm / Z + 1e-10f;
        const float f = mi / (Z * denom);
        float *o = out + ((size_t)g * H + i) * 64 + hd * 8;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = acc[d] * f;
    }
}

// LSTM pointwise (PyTorch gate order i,f,g,o).  gx row for (e,h) = gx[((e*S + t)*H + h)*256]; in_mask scales the input
// contribution ((x*m) W = m (x W) for m in {0,1}); blend (decode steps): h = h' m + h (1-m); post: h,c *= post_mask.
__global__ __launch_bounds__(64) void gst_lstm_pointwise_kernel(int E, int H, int S, int t, const float *__restrict__ gx,
                                                                const float *__restrict__ in_mask, const float *__restrict__ gh,
                                                                const float *__restrict__ b_ih, const float *__restrict__ b_hh,
                                                                float *__restrict__ h, float *__restrict__ c,
                                                                const float *__restrict__ blend_mask, const float *__restrict__ post_mask)
{
    const int n = blockIdx.x, d = threadIdx.x; // n = e*H + hh
    const int e = n / H, hh = n - e * H;
    const size_t r = ((size_t)e * S + t) * H + hh;
    const float m = in_mask[r];
    const float *gxr = gx + r * 256, *ghr = gh + (size_t)n * 256;
    const float gi = m * gxr[d] + b_ih[d] + ghr[d] + b_hh[d];
    const float gf = m * gxr[64 + d] + b_ih[64 + d] + ghr[64 + d] + b_hh[64 + d];
    const float gg = m * gxr[128 + d] + b_ih[128 + d] + ghr[128 + d] + b_hh[128 + d];
    const float go = m * gxr[192 + d] + b_ih[192 + d] + ghr[192 + d] + b_hh[192 + d];
    const float c0 = c[(size_t)n * 64 + d], h0 = h[(size_t)n * 64 + d];
    float cn = 1.0f / (1.0f + expf(-gf)) * c0 + 1.0f / (1.0f + expf(-gi)) * tanhf(gg);
    float hn = 1.0f / (1.0f + expf(-go)) * tanhf(cn);
    if (blend_mask) { const float bm = blend_mask[n]; hn = hn * bm + h0 * (1.0f - bm); cn = cn * bm + c0 * (1.0f - bm); }
    if (post_mask) { const float pm = post_mask[n]; hn *= pm; cn *= pm; }
    h[(size_t)n * 64 + d] = hn; c[(size_t)n * 64 + d] = cn;
}

// hidden2pos + raw2gaussian + the running sums of crowd_nav_interface_parallel.py:99-113 for decode step tt.
// acc[n][5] = running (mu_x, mu_y, sx^2, sy^2, corr*sx*sy).  One wavefront per pedestrian, lane = hidden unit.
__global__ __launch_bounds__(256) void gst_head_kernel(int N, int tt, const float *__restrict__ h, const float *__restrict__ W, const float *__restrict__ b,
                                                       const float *__restrict__ lm_fp, const float *__restrict__ last_pos, float *__restrict__ acc,
                                                       float *__restrict__ out_traj, float *__restrict__ x_sample)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float hv = h[(size_t)n * 64 + lane];
    float raw[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) raw[k] = wv_sum(hv * W[k * 64 + lane]) + b[k];
    if (lane == 0) {
        const float lm = lm_fp[n];
        const float sx = expf(raw[2]), sy = expf(raw[3]), corr = tanhf(raw[4]);
        float *a = acc + (size_t)n * 5;
        const float a0 = (tt ? a[0] : 0.f) + raw[0], a1 = (tt ? a[1] : 0.f) + raw[1];
        const float a2 = (tt ? a[2] : 0.f) + sx * sx, a3 = (tt ? a[3] : 0.f) + sy * sy, a4 = (tt ? a[4] : 0.f) + corr * sx * sy;
        a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3; a[4] = a4;
        const float sxc = sqrtf(a2), syc = sqrtf(a3);
        float *o = out_traj + ((size_t)n * GP + tt) * 5;
        o[0] = (a0 + last_pos[2 * n]) * lm + GST_INVALID * (1.0f - lm);
        o[1] = (a1 + last_pos[2 * n + 1]) * lm + GST_INVALID * (1.0f - lm);
        o[2] = sxc; o[3] = syc; o[4] = a4 / (sxc * syc);
        x_sample[2 * n] = raw[0] * lm; x_sample[2 * n + 1] = raw[1] * lm; // sampling = False: the mean, masked for the next step
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

constexpr size_t g_align(size_t x) { return (x + 255) & ~size_t(255); }

} // namespace

struct cn_gst {
    int H, maxE;
    bool weights_set;
    char *blob;
    float *emb_w, *emb_b, *in_w, *in_b, *out_w, *out_b, *n_w, *n_b, *n1_w, *n1_b, *l1_w, *l1_b, *l2_w, *l2_b;
    float *wih, *whh, *bih, *bhh, *h2p_w, *h2p_b;
    // workspace (rows = maxE * 5 * H)
    float *m_rel, *lm_fp, *rel, *last_pos, *x0, *qkv, *att, *x1, *x2, *ff, *xs, *gx, *gh, *h, *c, *acc, *x_sample;
    float *out_traj, *out_mask; // internal buffers for the wrapper path
    // VecPretextNormalize history
    float *ring_traj;  // [5][maxE][H][2]
    uint8_t *ring_mask; // [5][maxE][H]
    int ring_E, ring_pos;
};

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
    const size_t o_mrel = carve(R), o_lm = carve(N), o_rel = carve(2 * R), o_lp = carve(2 * N), o_x0 = carve(R * 64), o_qkv = carve(R * 192), o_att = carve(R * 64);
    const size_t o_x1 = carve(R * 64), o_x2 = carve(R * 64), o_ff = carve(R * 128), o_xs = carve(R * 64), o_gx = carve(R * 256), o_gh = carve(N * 256);
    const size_t o_h = carve(N * 64), o_c = carve(N * 64), o_acc = carve(N * 5), o_xsamp = carve(N * 2), o_ot = carve(N * GP * 5), o_om = carve(N);
    const size_t o_rt = carve((size_t)GT * N * 2), o_rm = carve(((size_t)GT * N + 3) / 4);
    char *base = nullptr;
    hipError_t herr = hipMalloc((void **)&base, off);
    if (herr != hipSuccess) { delete g; cn_set_error("cn_gst_create: hipMalloc(%zu) failed: %s", off, hipGetErrorString(herr)); return CN_ERR_HIP; }
    g->blob = base;
    auto F = [&](size_t o) { return (float *)(base + o); };
    float **wp[] = {&g->emb_w, &g->emb_b, &g->in_w, &g->in_b, &g->out_w, &g->out_b, &g->n_w, &g->n_b, &g->n1_w, &g->n1_b,
                    &g->l1_w, &g->l1_b, &g->l2_w, &g->l2_b, &g->wih, &g->whh, &g->bih, &g->bhh, &g->h2p_w, &g->h2p_b};
    for (int i = 0; i < 20; ++i) *wp[i] = F(o_w[i]);
    g->m_rel = F(o_mrel); g->lm_fp = F(o_lm); g->rel = F(o_rel); g->last_pos = F(o_lp); g->x0 = F(o_x0); g->qkv = F(o_qkv); g->att = F(o_att);
    g->x1 = F(o_x1); g->x2 = F(o_x2); g->ff = F(o_ff); g->xs = F(o_xs); g->gx = F(o_gx); g->gh = F(o_gh); g->h = F(o_h); g->c = F(o_c);
    g->acc = F(o_acc); g->x_sample = F(o_xsamp); g->out_traj = F(o_ot); g->out_mask = F(o_om);
    g->ring_traj = F(o_rt); g->ring_mask = (uint8_t *)(base + o_rm);
    g->ring_E = 0; g->ring_pos = 0; g->weights_set = false;
    *out = g;
    return CN_OK;
}

extern "C" int cn_gst_destroy(cn_gst *g)
{
    if (!g) return CN_OK;
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
    g->weights_set = true;
    return CN_OK;
}

// one NodeEncoderLayer over `rows` rows grouped in `groups` groups of H nodes; x2 [rows,2], mask [rows] -> g->xs [rows,64]
static int gst_transformer(cn_gst *g, int rows, int groups, const float *x2, const float *mask, hipStream_t st)
{
    const int H = g->H;
    int rc;
    hipLaunchKernelGGL(gst_embed_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, rows, x2, mask, g->emb_w, g->emb_b, g->n_w, g->n_b, g->x0);
    CN_CHECK_LAUNCH();
    const GemmBatch nb{0, 0, 0, 0, nullptr, 0};
    if ((rc = launch_gemm_t<128, 64, ACT_NONE>(rows, 192, 64, g->x0, 64, g->in_w, g->in_b, g->qkv, 192, st, nullptr, 1, nb, 1 << 30))) return rc;
    {
        const size_t per_wave = (size_t)(2 * H * 65 + 64) * sizeof(float);
        int wpb = (int)(65536 / per_wave); wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
        hipLaunchKernelGGL(gst_attention_kernel, dim3((groups + wpb - 1) / wpb), dim3(64 * wpb), per_wave * wpb, st, groups, H, g->qkv, mask, g->att);
        CN_CHECK_LAUNCH();
    }
    if ((rc = launch_gemm_t<128, 64, ACT_NONE>(rows, 64, 64, g->att, 64, g->out_w, g->out_b, g->x1, 64, st, nullptr, 1, GemmBatch{0, 0, 0, 0, g->x0, 64}, 1 << 30))) return rc;
    hipLaunchKernelGGL(gst_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, rows, g->x1, g->n1_w, g->n1_b, g->x2);
    CN_CHECK_LAUNCH();
    if ((rc = launch_gemm_t<128, 64, ACT_RELU>(rows, 128, 64, g->x2, 64, g->l1_w, g->l1_b, g->ff, 128, st, nullptr, 1, nb, 1 << 30))) return rc;
    if ((rc = launch_gemm_t<128, 64, ACT_NONE>(rows, 64, 128, g->ff, 128, g->l2_w, g->l2_b, g->xs, 64, st, nullptr, 1, GemmBatch{0, 0, 0, 0, g->x1, 64}, 1 << 30))) return rc;
    return CN_OK;
}

static int gst_forward(cn_gst *g, int E, const float *traj, long long se, long long sh, long long stt, const uint8_t *mask_u8, const float *mask_f,
                       long long me, long long mh, long long mt, int rot, float *out_traj, float *out_mask, hipStream_t st)
{
    if (!g->weights_set) { cn_set_error("cn_gst: call cn_gst_set_weights first"); return CN_ERR_STATE; }
    const int H = g->H, N = E * H, R = N * GT;
    int rc;
    const GemmBatch nb{0, 0, 0, 0, nullptr, 0};
    hipLaunchKernelGGL(gst_obs_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, st, E, H, traj, se, sh, stt, mask_u8, mask_f, me, mh, mt, rot,
                       g->m_rel, g->lm_fp, g->rel, g->last_pos);
    CN_CHECK_LAUNCH();
    // observation period: spatial encoding of all 5 slices at once, then the LSTM over time
    if ((rc = gst_transformer(g, R, E * GT, g->rel, g->m_rel, st))) return rc;
    if ((rc = launch_gemm_t<128, 64, ACT_NONE>(R, 256, 64, g->xs, 64, g->wih, nullptr, g->gx, 256, st, nullptr, 1, nb, 1 << 30))) return rc;
    CN_HIP(hipMemsetAsync(g->h, 0, (size_t)N * 64 * sizeof(float), st));
    CN_HIP(hipMemsetAsync(g->c, 0, (size_t)N * 64 * sizeof(float), st));
    for (int t = 0; t < GT; ++t) {
        if ((rc = launch_gemm_t<64, 64, ACT_NONE>(N, 256, 64, g->h, 64, g->whh, nullptr, g->gh, 256, st, nullptr, 1, nb, 1 << 30))) return rc;
        hipLaunchKernelGGL(gst_lstm_pointwise_kernel, dim3(N), dim3(64), 0, st, E, H, GT, t, g->gx, g->m_rel, g->gh, g->bih, g->bhh, g->h, g->c,
                           (const float *)nullptr, t == GT - 1 ? g->lm_fp : (const float *)nullptr);
        CN_CHECK_LAUNCH();
    }
    // prediction period (recursive decoding on the mean)
    for (int tt = 0; tt < GP; ++tt) {
        if (tt > 0) {
            if ((rc = gst_transformer(g, N, E, g->x_sample, g->lm_fp, st))) return rc;
            if ((rc = launch_gemm_t<128, 64, ACT_NONE>(N, 256, 64, g->xs, 64, g->wih, nullptr, g->gx, 256, st, nullptr, 1, nb, 1 << 30))) return rc;
            if ((rc = launch_gemm_t<64, 64, ACT_NONE>(N, 256, 64, g->h, 64, g->whh, nullptr, g->gh, 256, st, nullptr, 1, nb, 1 << 30))) return rc;
            hipLaunchKernelGGL(gst_lstm_pointwise_kernel, dim3(N), dim3(64), 0, st, E, H, 1, 0, g->gx, g->lm_fp, g->gh, g->bih, g->bhh, g->h, g->c,
                               g->lm_fp, (const float *)nullptr);
            CN_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(gst_head_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, tt, g->h, g->h2p_w, g->h2p_b, g->lm_fp, g->last_pos, g->acc, out_traj,
                           g->x_sample);
        CN_CHECK_LAUNCH();
    }
    if (out_mask != g->lm_fp) CN_HIP(hipMemcpyAsync(out_mask, g->lm_fp, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, st));
    return CN_OK;
}

extern "C" int cn_gst_predict(cn_gst *g, int E, const float *in_traj, const float *in_mask, float *out_traj, float *out_mask, void *stream)
{
    CN_REQUIRE(g && in_traj && in_mask && out_traj && out_mask && E >= 1 && E <= g->maxE, "cn_gst_predict: bad argument");
    const long long H = g->H;
    return gst_forward(g, E, in_traj, H * GT * 2, GT * 2, 2, nullptr, in_mask, H * GT, GT, 1, 0, out_traj, out_mask, (hipStream_t)stream);
}

extern "C" int cn_gst_wrapper_reset(cn_gst *g, int E, void *stream)
{
    CN_REQUIRE(g && E >= 1 && E <= g->maxE, "cn_gst_wrapper_reset: bad argument");
    const size_t n = (size_t)GT * E * g->H;
    hipLaunchKernelGGL(pretext_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, g->ring_traj, g->ring_mask);
    CN_CHECK_LAUNCH();
    g->ring_E = E; g->ring_pos = 0;
    return CN_OK;
}

extern "C" int cn_gst_wrapper_step(cn_gst *g, int E, const cn_obs *obs, float robot_plus_human_radius, float collision_penalty, float *rewards,
                                   float *spatial_edges_out, void *stream)
{
    CN_REQUIRE(g && obs && obs->robot_node && obs->spatial_edges && obs->visible_masks && spatial_edges_out, "cn_gst_wrapper_step: null argument");
    if (g->ring_E != E) { cn_set_error("cn_gst_wrapper_step: call cn_gst_wrapper_reset(E=%d) first", E); return CN_ERR_STATE; }
    hipStream_t st = (hipStream_t)stream;
    const int H = g->H, D = 2 * (GP + 1), N = E * H;
    // deque.append: the oldest slot is overwritten, time order = (pos+1 .. pos+5) % 5
    hipLaunchKernelGGL(pretext_push_kernel, dim3((N + 255) / 256), dim3(256), 0, st, E, H, D, obs->robot_node, obs->spatial_edges, obs->visible_masks,
                       g->ring_traj, g->ring_mask, g->ring_pos);
    CN_CHECK_LAUNCH();
    const int rot = (g->ring_pos + 1) % GT;
    g->ring_pos = rot;
    if (int rc = gst_forward(g, E, g->ring_traj, (long long)H * 2, 2, (long long)N * 2, g->ring_mask, nullptr, H, 1, N, rot, g->out_traj, g->lm_fp, st)) return rc;
    float *rw = rewards;
    hipLaunchKernelGGL(pretext_post_kernel, dim3(E), dim3(64), 0, st, E, H, D, obs->robot_node, obs->spatial_edges, g->out_traj, g->lm_fp,
                       robot_plus_human_radius, collision_penalty, rw ? rw : g->acc /*scratch*/, spatial_edges_out);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
