// policy.hip -- rollout-time forward of the attention-interaction-graph policy (selfAttn_merge_srnn + DiagGaussian)
// on gfx950.  Reference: rl/networks/selfAttn_srnn_temp_node.py:360-449, rl/networks/model.py:56-80.
//
// Pipeline per call (E envs, H humans, M = E*H rows), all on the caller's stream, no host sync:
//   embed0        [M,D]   -> [M,128]  ReLU                      (K = 2 or 12: VALU)
//   gemm          [M,128] -> [M,512]  ReLU                      embedding_layer.2
//   gemm          [M,512] -> [M,1536]                           folded (q|k|v)_linear ∘ in_proj, 1/sqrt(64) folded into q
//   hh_attention  per (env, head): softmax(QK^T + key padding mask) V  -> [M,512]
//   gemm          [M,512] -> [M,256]  ReLU                      folded out_proj ∘ spatial_linear
//   robot_embed   [E,9]   -> [E,256]  ReLU ; gemm -> [E,256]    robot_linear, u = spatial_edge_layer^T temporal_edge_layer(.)
//   hr_attention  per env: scores u . out_sp_j, masked softmax over humans, weighted sum of [H,256]
//   gemms + gru_pointwise + gemms(tanh) + gauss_head            EndRNN, actor/critic, DiagGaussian
// The reference computes in fp32 and the parity bar is 1e-4, which plain bf16 inputs cannot hold at K = 512.  The three
// large products (embedding_layer.2, q|k|v, out_proj∘spatial_linear) therefore run as bf16x3 split-precision MFMA
// (gemm3.h: hi/lo bf16 pairs, three v_mfma_f32_32x32x16_bf16 per term, fp32 accumulate, ~2e-5 from fp32; default) or as
// exact fp32 MFMA (cn_policy_set_gemm_mode(p, 0)); every other product is exact fp32 on v_mfma_f32_32x32x2_f32 (gemm.h).
// The rows of the human-human block are the compacted "live" (env, human) rows only, the robot-node launches run on a side
// stream beside that block.
// The two affine pairs without a nonlinearity in between are folded once per weight snapshot (fp64 accumulation),
// which removes 4 of the 9 [M,512]x[512,512] products (83.65 -> 52.2 MFLOP per env-step at H = 20).
#include "common.h"
#include "gemm.h"
#include "gemm3.h"
#include "hh_fused.h"
#include "rn_fused.h"
#include "row_plan.h"
#include "train_internal.h"

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <new>
#include <vector>

namespace {

// Row compaction: row_off[e] = sum_{e' < e} nd(e'), nd = clamp(detected_human_num, 1, H); row_off[E] = number of live
// (env, human) rows.  Padded humans (index >= nd) only ever meet an exactly-zero robot-human attention weight, so the
// whole human-human block runs on live rows only.  Single block, Hillis-Steele scan over per-thread chunk sums.
// cls_cnt[2] / cls_list[2][E] (optional): the envs of the two rare big attention size classes (16 < nd <= 32, nd > 32);
// the order inside a bin is arbitrary (LDS atomics) and has no effect on any result (a unit writes only its own rows).
__global__ __launch_bounds__(1024) void row_offsets_kernel(int E, int H, const float *__restrict__ det, int *__restrict__ row_off,
                                                           unsigned long long *__restrict__ live_total, int *__restrict__ cls_cnt,
                                                           int *__restrict__ cls_list)
{
    __shared__ int part[1024];
    __shared__ int cnt[2];
    const int t = threadIdx.x;
    if (t < 2) cnt[t] = 0;
    const int chunk = (E + 1023) / 1024;
    const int lo = t * chunk, hi = min(lo + chunk, E);
    int sum = 0;
    for (int e = lo; e < hi; ++e) { int nd = (int)det[e]; nd = nd < 1 ? 1 : (nd > H ? H : nd); sum += nd; }
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum; // exclusive prefix of this thread's chunk
    for (int e = lo; e < hi; ++e) {
        row_off[e] = run;
        int nd = (int)det[e]; nd = nd < 1 ? 1 : (nd > H ? H : nd);
        run += nd;
        if (cls_list && nd > 16) {
            const int c = nd <= 32 ? 0 : 1;
            cls_list[(size_t)c * E + atomicAdd(&cnt[c], 1)] = e;
        }
    }
    if (cls_cnt) {
        __syncthreads();
        if (t < 2) cls_cnt[t] = cnt[t];
    }
    if (t == 1023) {
        row_off[E] = part[1023];
        if (live_total) *live_total += (unsigned long long)part[1023]; // measurement aid: total live rows over the profiled launches
    }
}

// embedding_layer.0 (K = D <= 16) on live rows: out[row_off[e] + j][n] = relu(sum_d x[e][j][d] * W[n][d] + b[n]), n < 128
__global__ __launch_bounds__(128) void embed0_kernel(int E, int H, int D, const float *__restrict__ x, const float *__restrict__ W,
                                                     const float *__restrict__ b, const int *__restrict__ row_off, float *__restrict__ out)
{
    const int n = threadIdx.x;
    float w[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) w[d] = d < D ? W[n * D + d] : 0.0f;
    const float bn = b[n];
    for (int e = blockIdx.x; e < E; e += gridDim.x) {
        const int r0 = row_off[e], nd = row_off[e + 1] - r0;
        for (int j = 0; j < nd; ++j) {
            const float *xr = x + ((size_t)e * H + j) * D;
            float acc = bn;
#pragma unroll
            for (int d = 0; d < 16; ++d)
                if (d < D) acc += xr[d] * w[d];
            out[(size_t)(r0 + j) * 128 + n] = fmaxf(acc, 0.0f);
        }
    }
}

// robot_linear.0: out[e][n] = relu(W[n][0:2] . temporal_edges[e] + W[n][2:9] . robot_node[e] + b[n]), n < 256
// (torch.cat((temporal_edges, robot_node), -1), selfAttn_srnn_temp_node.py:397)
__global__ __launch_bounds__(256) void robot_embed_kernel(int E, const float *__restrict__ temporal, const float *__restrict__ robot_node,
                                                          const float *__restrict__ W, const float *__restrict__ b, float *__restrict__ out)
{
    const int n = threadIdx.x;
    float w[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) w[d] = W[n * 9 + d];
    const float bn = b[n];
    for (int e = blockIdx.x; e < E; e += gridDim.x) {
        float acc = bn;
        acc += temporal[e * 2] * w[0];
        acc += temporal[e * 2 + 1] * w[1];
#pragma unroll
        for (int d = 0; d < 7; ++d) acc += robot_node[e * 7 + d] * w[2 + d];
        out[(size_t)e * 256 + n] = fmaxf(acc, 0.0f);
    }
}

// Human-human multi-head attention core (torch.nn.MultiheadAttention with key_padding_mask, 8 heads x 64) on the
// compacted rows: one wavefront per (env, head).
//   load    : lane d reads element d of every live row: Q, K rows go to LDS (row stride 68 floats = 16-byte aligned and
//             conflict-free for ds_read_b128), V stays in registers (lane d only ever needs column d of V)
//   scores  : lanes enumerate (query i, key j) pairs, 64 pairs per pass; each dot product is 16 x (2 b128 reads + 4 FMA)
//   softmax : lane i owns row i of S (LDS, row stride CAP, zero padded so P*V can read float4s)
//   P*V     : lane d: o[i][d] = sum_j S[i][j] * v[j]
// Masked keys are simply absent (softmax over the nd live keys == softmax with -inf on the padded ones).
// CAP in {8,16,32,64} are size classes sharing one launch grid: class (cap_lo, CAP] handles the units with that many
// detected humans (the common case of ~6 uses 4.6 KB of LDS per wavefront -> high occupancy); other units exit at once.
template <int CAP>
__global__ __launch_bounds__(256) void hh_attention_kernel(int E, int cap_lo, const float *__restrict__ qkv, const int *__restrict__ row_off,
                                                           const int *__restrict__ cls_cnt, const int *__restrict__ cls_list,
                                                           float *__restrict__ out, float scale)
{
    constexpr int RS = 68;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    // wavefronts walk (env, head) units.  Without a class list every unit is inspected in env order and filtered by nd
    // (grid = one unit per wavefront); with one (the rare big classes, whose 39-65 KB blocks would otherwise queue up
    // just to exit) a resident-sized grid walks only that class's envs.
    const int n_units = (cls_list ? *cls_cnt : E) * 8;
    for (int unit = blockIdx.x * wpb + wave; unit < n_units; unit += gridDim.x * wpb) {
    const int e = cls_list ? cls_list[unit >> 3] : unit >> 3, head = unit & 7;
    const int r0 = row_off[e], nd = row_off[e + 1] - r0;
    if (nd <= cap_lo || nd > CAP) continue; // another size class handles this unit
    float *Ks = smem + (size_t)wave * (2 * CAP * RS + CAP * CAP);
    float *Qs = Ks + CAP * RS;
    float *S = Qs + CAP * RS;
    const float *base = qkv + (size_t)r0 * 1536 + head * 64 + lane;
    float v[CAP];
#pragma unroll
    for (int j0 = 0; j0 < CAP; j0 += 8) {
        if (j0 < nd) { // wave-uniform
            float q[8], k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float *row = base + (size_t)(j0 + u < nd ? j0 + u : 0) * 1536;
                q[u] = row[0]; k[u] = row[512]; v[j0 + u] = row[1024];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < nd) { Qs[(j0 + u) * RS + lane] = q[u]; Ks[(j0 + u) * RS + lane] = k[u]; }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[j0 + u] = 0.0f;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): LDS writes of this wavefront visible to its own reads
    const int npairs = nd * nd;
    for (int q = lane; q < npairs; q += 64) {
        const int qi = q / nd, qj = q - qi * nd;
        const f32x4 *qp = reinterpret_cast<const f32x4 *>(Qs + qi * RS);
        const f32x4 *kp = reinterpret_cast<const f32x4 *>(Ks + qj * RS);
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const f32x4 a = qp[d], b = kp[d];
            s += a[0] * b[0]; s += a[1] * b[1]; s += a[2] * b[2]; s += a[3] * b[3];
        }
        S[qi * CAP + qj] = s * scale;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane < nd) {
        // whole row in registers (16-byte reads, all issued before the first use): as loops over the run-time nd, every LDS read
        // waited for the one before it.  Entries past nd hold stale scores of earlier units: masked here, stored as zeros.
        float *row = S + lane * CAP;
        float p[CAP];
#pragma unroll
        for (int j4 = 0; j4 < CAP / 4; ++j4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(row + 4 * j4);
#pragma unroll
            for (int u = 0; u < 4; ++u) p[4 * j4 + u] = a[u];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < CAP; ++j) mx = j < nd ? fmaxf(mx, p[j]) : mx;
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < CAP; ++j) { p[j] = j < nd ? expf(p[j] - mx) : 0.0f; sum += p[j]; }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int j4 = 0; j4 < CAP / 4; ++j4)
            *reinterpret_cast<f32x4 *>(row + 4 * j4) = f32x4{p[4 * j4] * inv, p[4 * j4 + 1] * inv, p[4 * j4 + 2] * inv, p[4 * j4 + 3] * inv};
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int i = 0; i < nd; ++i) {
        const f32x4 *prow = reinterpret_cast<const f32x4 *>(S + i * CAP);
        float o = 0.0f;
#pragma unroll
        for (int j4 = 0; j4 < CAP / 4; ++j4) {
            const f32x4 p = prow[j4];
            o += p[0] * v[4 * j4]; o += p[1] * v[4 * j4 + 1]; o += p[2] * v[4 * j4 + 2]; o += p[3] * v[4 * j4 + 3];
        }
        out[(size_t)(r0 + i) * 512 + head * 64 + lane] = o;
    }
    __builtin_amdgcn_wave_barrier(); // the next unit reuses this wavefront's LDS slices
    }
}

template <int CAP>
static int launch_hh_attention(int E, int cap_lo, const float *qkv, const int *row_off, float *out, hipStream_t st, float scale = 1.0f,
                               const int *cls_cnt = nullptr, const int *cls_list = nullptr)
{
    const size_t per_wave = (size_t)(2 * CAP * 68 + CAP * CAP) * sizeof(float);
    int wpb = (int)(65536 / per_wave); wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
    int per_cu = (int)((160 * 1024) / (per_wave * wpb)); per_cu = per_cu > 8 ? 8 : per_cu; // resident blocks per CU (LDS / 32-wave cap)
    int blocks = (E * 8 + wpb - 1) / wpb;
    if (cls_list && blocks > 256 * per_cu) blocks = 256 * per_cu;
    hipLaunchKernelGGL(hh_attention_kernel<CAP>, dim3(blocks), dim3(64 * wpb), per_wave * wpb, st, E, cap_lo, qkv, row_off, cls_cnt, cls_list, out, scale);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// Size classes of the (sample, head) units for the training kernels: class c holds the samples with cap_lo(c) < nd <= cap(c),
// caps 8 / 16 / 32 / 64.  cls = [4] counts followed by [4][B] sample lists (order inside a list is arbitrary -- a unit writes
// only its own rows).  One launch per class then walks exactly its own units with an LDS footprint sized for that class.
__global__ __launch_bounds__(256) void hh_classify_kernel(int B, const int *__restrict__ row_off, int *__restrict__ cls)
{
    __shared__ int cnt[4], base[4];
    const int t = threadIdx.x;
    if (t < 4) cnt[t] = 0;
    __syncthreads();
    const int b = blockIdx.x * blockDim.x + t;
    int c = -1, slot = 0;
    if (b < B) {
        const int nd = row_off[b + 1] - row_off[b];
        c = nd <= 8 ? 0 : (nd <= 16 ? 1 : (nd <= 32 ? 2 : 3));
        slot = atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    if (t < 4) base[t] = cnt[t] ? atomicAdd(&cls[t], cnt[t]) : 0;
    __syncthreads();
    if (c >= 0) cls[4 + (size_t)c * B + base[c] + slot] = b;
}

// Backward of the attention core for training (PPO update): per (sample, head) on the compacted rows.
//   S = scale * Q K^T, P = softmax(S), O = P V ;   given dO:
//   dV = P^T dO ; dP = dO V^T ; dS = scale * P .* (dP - rowsum(dP .* P)) ; dQ = dS K ; dK = dS^T Q
// One wavefront per unit; Q, K, V, dO rows in LDS (stride 68), P and dS as nd x nd matrices (stride CAP) in LDS.  CAP is the size
// class (see hh_classify_kernel): the common class of <= 8 detected humans needs 9 KB per wavefront instead of the 25 KB of H = 20,
// so 16 wavefronts are resident per CU instead of 6.
template <int CAP>
__global__ __launch_bounds__(256) void hh_attention_bwd_kernel(int B, const float *__restrict__ qkv, const int *__restrict__ row_off,
                                                               const int *__restrict__ cls_cnt, const int *__restrict__ cls_list,
                                                               const float *__restrict__ d_out, float *__restrict__ d_qkv, float scale)
{
    constexpr int RS = 68;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int n_units = (cls_list ? *cls_cnt : B) * 8;
    constexpr bool REG = CAP <= 32; // softmax rows and the K / Q / dO columns in registers, see below
    float *Qs = smem + (size_t)wave * (4 * CAP * RS + (REG ? 4 : 2) * CAP * CAP);
    float *Ks = Qs + CAP * RS, *Vs = Ks + CAP * RS, *Gs = Vs + CAP * RS; // Gs = dO rows
    float *P = Gs + CAP * RS, *dS = P + CAP * CAP;
    float *PT = dS + CAP * CAP, *dST = PT + CAP * CAP; // (REG only) transposed copies: dK and dV walk columns of dS and P
    if (REG) { // rows at or past nd are read (with zero weights) by the unrolled loops below: they must hold finite numbers
        for (int x = lane; x < 4 * CAP * RS; x += 64) Qs[x] = 0.0f;
    }
    // A wavefront walks ~240 units, and a unit starts with three DEPENDENT memory round trips (class list -> row offsets -> rows)
    // before any arithmetic.  For the small classes (4 CAP registers) the walk is a three-stage pipeline instead: while unit u is
    // computed, the rows of unit u+1, the row offsets of unit u+2 and the sample id of unit u+3 are in flight.
    constexpr bool PIPE = CAP <= 16;
    float pq[PIPE ? CAP : 1], pk[PIPE ? CAP : 1], pv[PIPE ? CAP : 1], pg[PIPE ? CAP : 1];
    const int stride = gridDim.x * wpb;
    int unit = blockIdx.x * wpb + wave;
    auto sample_of = [&](int u) { return u < n_units ? (cls_list ? cls_list[u >> 3] : u >> 3) : 0; };
    auto request_rows = [&](int r0n, int ndn, int head) {
        const float *base = qkv + (size_t)r0n * 1536 + head * 64 + lane;
        const float *gbase = d_out + (size_t)r0n * 512 + head * 64 + lane;
#pragma unroll
        for (int j = 0; j < (PIPE ? CAP : 0); ++j)
            if (j < ndn) { // wave-uniform
                pq[j] = base[(size_t)j * 1536]; pk[j] = base[(size_t)j * 1536 + 512]; pv[j] = base[(size_t)j * 1536 + 1024];
                pg[j] = gbase[(size_t)j * 512];
            }
    };
    int c_r0 = 0, c_nd = 0, n_lo = 0, n_hi = 0, b2 = 0; // rows in flight belong to (c_r0, c_nd); raw offsets of the unit after it; sample after that
    if (PIPE && unit < n_units) {
        const int b0 = sample_of(unit), b1 = sample_of(unit + stride);
        b2 = sample_of(unit + 2 * stride);
        c_r0 = row_off[b0]; c_nd = row_off[b0 + 1] - c_r0;
        n_lo = row_off[b1]; n_hi = row_off[b1 + 1];
        request_rows(c_r0, c_nd, unit & 7);
    }
    for (; unit < n_units; unit += stride) {
        const int head = unit & 7;
        int r0, nd;
        if (PIPE) {
            r0 = c_r0; nd = c_nd;
#pragma unroll
            for (int j = 0; j < (PIPE ? CAP : 0); ++j)
                if (j < nd && nd <= CAP) { Qs[j * RS + lane] = pq[j]; Ks[j * RS + lane] = pk[j]; Vs[j * RS + lane] = pv[j]; Gs[j * RS + lane] = pg[j]; }
            c_r0 = n_lo; c_nd = n_hi - n_lo;
            request_rows(c_r0, c_nd, (unit + stride) & 7);        // (past the end: sample 0 again, never used)
            n_lo = row_off[b2]; n_hi = row_off[b2 + 1];
            b2 = sample_of(unit + 3 * stride);
            if (nd > CAP) continue; // (only without a class list: another launch handles it)
        } else {
            const int b = cls_list ? cls_list[unit >> 3] : unit >> 3;
            r0 = row_off[b]; nd = row_off[b + 1] - r0;
            if (nd > CAP) continue; // (only without a class list: another launch handles it)
            const float *base = qkv + (size_t)r0 * 1536 + head * 64 + lane;
            const float *gbase = d_out + (size_t)r0 * 512 + head * 64 + lane;
#pragma unroll 4
            for (int j = 0; j < nd; ++j) {
                const float q = base[(size_t)j * 1536], k = base[(size_t)j * 1536 + 512], v = base[(size_t)j * 1536 + 1024], g = gbase[(size_t)j * 512];
                Qs[j * RS + lane] = q; Ks[j * RS + lane] = k; Vs[j * RS + lane] = v; Gs[j * RS + lane] = g;
            }
        }
        if (REG) { // entries outside nd x nd stay zero for this unit
#pragma unroll
            for (int x = 0; x < 4 * CAP * CAP; x += 64) P[x + lane] = 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int npairs = nd * nd;
        for (int q = lane; q < npairs; q += 64) {
            const int qi = q / nd, qj = q - qi * nd;
            const f32x4 *qp = reinterpret_cast<const f32x4 *>(Qs + qi * RS), *kp = reinterpret_cast<const f32x4 *>(Ks + qj * RS);
            const f32x4 *gp = reinterpret_cast<const f32x4 *>(Gs + qi * RS), *vp = reinterpret_cast<const f32x4 *>(Vs + qj * RS);
            float s = 0.0f, dp = 0.0f;
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                const f32x4 a = qp[d], bb = kp[d], g = gp[d], v = vp[d];
                s += a[0] * bb[0]; s += a[1] * bb[1]; s += a[2] * bb[2]; s += a[3] * bb[3];
                dp += g[0] * v[0]; dp += g[1] * v[1]; dp += g[2] * v[2]; dp += g[3] * v[3];
            }
            P[qi * CAP + qj] = s * scale;
            dS[qi * CAP + qj] = dp;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        float *ob = d_qkv + (size_t)r0 * 1536 + head * 64 + lane;
        if (REG) {
            // The loops over nd below have run-time bounds: left as loops, every LDS read waits for the one before it (a (sample,
            // head) unit cost ~nd^2 serial LDS round trips).  Here they run to the compile-time CAP on whole rows in registers
            // (16-byte reads, all issued before the first use); entries past nd are zeros, so they add nothing.
            if (lane < nd) {
                float p[CAP], d[CAP];
#pragma unroll
                for (int j4 = 0; j4 < CAP / 4; ++j4) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(P + lane * CAP + 4 * j4), b = *reinterpret_cast<const f32x4 *>(dS + lane * CAP + 4 * j4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { p[4 * j4 + u] = a[u]; d[4 * j4 + u] = b[u]; }
                }
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < CAP; ++j) mx = j < nd ? fmaxf(mx, p[j]) : mx;
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < CAP; ++j) { p[j] = j < nd ? expf(p[j] - mx) : 0.0f; sum += p[j]; }
                const float inv = 1.0f / sum;
                float rd = 0.0f;
#pragma unroll
                for (int j = 0; j < CAP; ++j) { p[j] *= inv; rd += d[j] * p[j]; }
#pragma unroll
                for (int j = 0; j < CAP; ++j) d[j] = scale * p[j] * (d[j] - rd);
#pragma unroll
                for (int j4 = 0; j4 < CAP / 4; ++j4)
                    *reinterpret_cast<f32x4 *>(dS + lane * CAP + 4 * j4) = f32x4{d[4 * j4], d[4 * j4 + 1], d[4 * j4 + 2], d[4 * j4 + 3]};
#pragma unroll
                for (int j = 0; j < CAP; ++j) { PT[j * CAP + lane] = p[j]; dST[j * CAP + lane] = d[j]; }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            float kr[CAP], qr[CAP], gr[CAP]; // column `lane` of K, Q, dO
#pragma unroll
            for (int i = 0; i < CAP; ++i) { kr[i] = Ks[i * RS + lane]; qr[i] = Qs[i * RS + lane]; gr[i] = Gs[i * RS + lane]; }
            for (int j = 0; j < nd; ++j) {
                float dq = 0.0f, dk = 0.0f, dv = 0.0f; // row j of dQ, dK, dV, column `lane`
#pragma unroll
                for (int i4 = 0; i4 < CAP / 4; ++i4) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(dS + j * CAP + 4 * i4), b = *reinterpret_cast<const f32x4 *>(dST + j * CAP + 4 * i4);
                    const f32x4 c = *reinterpret_cast<const f32x4 *>(PT + j * CAP + 4 * i4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { dq += a[u] * kr[4 * i4 + u]; dk += b[u] * qr[4 * i4 + u]; dv += c[u] * gr[4 * i4 + u]; }
                }
                ob[(size_t)j * 1536] = dq; ob[(size_t)j * 1536 + 512] = dk; ob[(size_t)j * 1536 + 1024] = dv;
            }
        } else {
        if (lane < nd) {
            float *prow = P + lane * CAP, *drow = dS + lane * CAP;
            float mx = -INFINITY;
            for (int j = 0; j < nd; ++j) mx = fmaxf(mx, prow[j]);
            float sum = 0.0f;
            for (int j = 0; j < nd; ++j) { const float e = expf(prow[j] - mx); prow[j] = e; sum += e; }
            const float inv = 1.0f / sum;
            float rd = 0.0f;
            for (int j = 0; j < nd; ++j) { prow[j] *= inv; rd += drow[j] * prow[j]; }
            for (int j = 0; j < nd; ++j) drow[j] = scale * prow[j] * (drow[j] - rd);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int j = 0; j < nd; ++j) {
            float dq = 0.0f, dk = 0.0f, dv = 0.0f; // row j of dQ, dK, dV, column `lane`
            for (int i = 0; i < nd; ++i) {
                dq += dS[j * CAP + i] * Ks[i * RS + lane];
                dk += dS[i * CAP + j] * Qs[i * RS + lane];
                dv += P[i * CAP + j] * Gs[i * RS + lane];
            }
            ob[(size_t)j * 1536] = dq; ob[(size_t)j * 1536 + 512] = dk; ob[(size_t)j * 1536 + 1024] = dv;
        }
        }
        __builtin_amdgcn_wave_barrier(); // the next unit reuses this wavefront's LDS slices
    }
}

// The classes of 9 .. 16 and 17 .. 32 detected humans on the matrix pipe (exact fp32: v_mfma_f32_16x16x4_f32).  One wavefront per (sample, head) unit as
// above, the unit's five small products as 16 x 16 MFMA tiles instead of ~nd^2 dependent LDS reads and FMAs per lane:
//   S = Q K^T, dP = dO V^T            A / B = 16 consecutive features of row (lane & 15), feature block lane >> 4: four 16-byte LDS reads each
//   softmax / dS                      in the accumulator layout (lane holds rows 4 (lane >> 4) + r of column lane & 15): row sums by DPP rotations
//   dV = P^T dO, dK = dS^T Q          A = the accumulator registers themselves (P[4 kb + s][lane & 15] IS register s of this lane)
//   dQ = dS K                         A = dS read back transposed from a 16 x 17 LDS tile
// With the k index of a product taken as (lane >> 4, step) -> 4 (lane >> 4) + step (same permutation for A and B, so the sums are unchanged).
// Rows at or past nd hold stale finite data of earlier units: P is masked to zero there, which zeroes every term they appear in.
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int CTRL>
__device__ __forceinline__ float row_rot(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
__device__ __forceinline__ float row16_sum(float v) { v += row_rot<0x128>(v); v += row_rot<0x124>(v); v += row_rot<0x122>(v); v += row_rot<0x121>(v); return v; }
__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, row_rot<0x128>(v)); v = fmaxf(v, row_rot<0x124>(v)); v = fmaxf(v, row_rot<0x122>(v)); v = fmaxf(v, row_rot<0x121>(v));
    return v;
}
template <int NT> // NT x NT tiles of 16 x 16: classes of <= 16 (NT = 1) and <= 32 (NT = 2) detected humans
__global__ __launch_bounds__(256) void hh_attention_bwd_mfma_kernel(int B, const float *__restrict__ qkv, const int *__restrict__ row_off,
                                                                    const int *__restrict__ cls_cnt, const int *__restrict__ cls_list,
                                                                    const float *__restrict__ d_out, float *__restrict__ d_qkv, float scale)
{
    constexpr int CAP = 16 * NT, RS = 68, TS = CAP + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int n_units = (cls_list ? *cls_cnt : B) * 8;
    float *Qs = smem + (size_t)wave * (4 * CAP * RS + CAP * TS);
    float *Ks = Qs + CAP * RS, *Vs = Ks + CAP * RS, *Gs = Vs + CAP * RS; // Gs = dO rows
    float *dSt = Gs + CAP * RS;                                           // dS, CAP x (CAP + 1)
    for (int x = lane; x < 4 * CAP * RS; x += 64) Qs[x] = 0.0f;
    const int l15 = lane & 15, kb = lane >> 4;
    float pq[CAP], pk[CAP], pv[CAP], pg[CAP];
    const int stride = gridDim.x * wpb;
    int unit = blockIdx.x * wpb + wave;
    auto sample_of = [&](int u) { return u < n_units ? (cls_list ? cls_list[u >> 3] : u >> 3) : 0; };
    auto request_rows = [&](int r0n, int ndn, int head) {
        const float *base = qkv + (size_t)r0n * 1536 + head * 64 + lane;
        const float *gbase = d_out + (size_t)r0n * 512 + head * 64 + lane;
#pragma unroll
        for (int j = 0; j < CAP; ++j)
            if (j < ndn) { // wave-uniform
                pq[j] = base[(size_t)j * 1536]; pk[j] = base[(size_t)j * 1536 + 512]; pv[j] = base[(size_t)j * 1536 + 1024];
                pg[j] = gbase[(size_t)j * 512];
            }
    };
    int c_r0 = 0, c_nd = 0, n_lo = 0, n_hi = 0, b2 = 0;
    if (unit < n_units) {
        const int b0 = sample_of(unit), b1 = sample_of(unit + stride);
        b2 = sample_of(unit + 2 * stride);
        c_r0 = row_off[b0]; c_nd = row_off[b0 + 1] - c_r0;
        n_lo = row_off[b1]; n_hi = row_off[b1 + 1];
        request_rows(c_r0, c_nd, unit & 7);
    }
    for (; unit < n_units; unit += stride) {
        const int head = unit & 7;
        const int r0 = c_r0, nd = c_nd;
#pragma unroll
        for (int j = 0; j < CAP; ++j)
            if (j < nd && nd <= CAP) { Qs[j * RS + lane] = pq[j]; Ks[j * RS + lane] = pk[j]; Vs[j * RS + lane] = pv[j]; Gs[j * RS + lane] = pg[j]; }
        c_r0 = n_lo; c_nd = n_hi - n_lo;
        request_rows(c_r0, c_nd, (unit + stride) & 7);        // (past the end: sample 0 again, never used)
        n_lo = row_off[b2]; n_hi = row_off[b2 + 1];
        b2 = sample_of(unit + 3 * stride);
        if (nd > CAP) continue; // (only without a class list: another launch handles it)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---- S = Q K^T and dP = dO V^T: tile (ti, tj) = rows 16 ti .., columns 16 tj .. ----
        f32x4m S[NT][NT], dP[NT][NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) { S[ti][tj] = f32x4m{0.f, 0.f, 0.f, 0.f}; dP[ti][tj] = f32x4m{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4m q4[NT], k4[NT], g4[NT], v4[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int o = (16 * tt + l15) * RS + 16 * kb + 4 * t;
                q4[tt] = *reinterpret_cast<const f32x4m *>(Qs + o); k4[tt] = *reinterpret_cast<const f32x4m *>(Ks + o);
                g4[tt] = *reinterpret_cast<const f32x4m *>(Gs + o); v4[tt] = *reinterpret_cast<const f32x4m *>(Vs + o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        S[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[ti][u], k4[tj][u], S[ti][tj], 0, 0, 0);
                        dP[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(g4[ti][u], v4[tj][u], dP[ti][tj], 0, 0, 0);
                    }
        }
        // ---- softmax over the keys (column 16 tj + lane & 15) of every query row 16 ti + 4 kb + r; dS = scale * P * (dP - sum_j dP P) ----
        f32x4m P[NT][NT], dS[NT][NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool rowv = 16 * ti + 4 * kb + r < nd;
                float sv[NT], mx = -INFINITY;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) { sv[tj] = (rowv && 16 * tj + l15 < nd) ? S[ti][tj][r] * scale : -INFINITY; mx = fmaxf(mx, sv[tj]); }
                mx = row16_max(mx);
                float e[NT], sum = 0.0f;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) { e[tj] = (rowv && 16 * tj + l15 < nd) ? expf(sv[tj] - mx) : 0.0f; sum += e[tj]; }
                sum = row16_sum(sum);
                const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
                float rd = 0.0f;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) { e[tj] *= inv; rd += dP[ti][tj][r] * e[tj]; } // (e = 0 where masked)
                rd = row16_sum(rd);
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    const float d = scale * e[tj] * (dP[ti][tj][r] - rd);
                    P[ti][tj][r] = e[tj];
                    dS[ti][tj][r] = d;
                    dSt[(16 * ti + 4 * kb + r) * TS + 16 * tj + l15] = d;
                }
            }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // ---- dV = P^T dO, dK = dS^T Q (rows = keys 16 tj ..), dQ = dS K (rows = queries 16 ti ..): four 16-feature column blocks each ----
        float *ob = d_qkv + (size_t)r0 * 1536 + head * 64 + l15;
#pragma unroll
        for (int to = 0; to < NT; ++to) {   // output row tile
            float dsT[NT][4];               // dS[16 to + lane & 15][16 tj + 4 kb + s]
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) dsT[tj][sI] = dSt[(16 * to + l15) * TS + 16 * tj + 4 * kb + sI];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x4m aq = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tc = 0; tc < NT; ++tc)  // contraction tile
#pragma unroll
                    for (int sI = 0; sI < 4; ++sI) {
                        const int ro = (16 * tc + 4 * kb + sI) * RS + 16 * cb + l15;
                        av = __builtin_amdgcn_mfma_f32_16x16x4f32(P[tc][to][sI], Gs[ro], av, 0, 0, 0);
                        ak = __builtin_amdgcn_mfma_f32_16x16x4f32(dS[tc][to][sI], Qs[ro], ak, 0, 0, 0);
                        aq = __builtin_amdgcn_mfma_f32_16x16x4f32(dsT[tc][sI], Ks[ro], aq, 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * to + 4 * kb + r;
                    if (row < nd) {
                        float *o = ob + (size_t)row * 1536 + 16 * cb;
                        o[0] = aq[r]; o[512] = ak[r]; o[1024] = av[r];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier(); // the next unit reuses this wavefront's LDS slices
    }
}

template <int CAP>
static int launch_hh_attention_bwd(int B, const float *qkv, const int *row_off, const int *cls, int c, const float *d_out, float *d_qkv, float scale,
                                   hipStream_t st)
{
    if (CAP == 16 || CAP == 32) {
        constexpr int NT = CAP == 32 ? 2 : 1;
        const size_t per_wave = (size_t)(4 * CAP * 68 + CAP * (CAP + 1)) * sizeof(float); // 18.5 KB / 39 KB
        const int wpb = CAP == 16 ? 4 : 2, per_cu = 2;                                    // 8 / 4 wavefronts per CU
        int blocks = (B * 8 + wpb - 1) / wpb;
        if (blocks > 256 * per_cu) blocks = 256 * per_cu;
        static CnLdsOptIn opt_in; // per device
        int opt_dev;
        if (opt_in.needed(&opt_dev)) {
            CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hh_attention_bwd_mfma_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            opt_in.done(opt_dev);
        }
        hipLaunchKernelGGL(hh_attention_bwd_mfma_kernel<NT>, dim3(blocks), dim3(64 * wpb), per_wave * wpb, st, B, qkv, row_off, cls + c, cls + 4 + (size_t)c * B,
                           d_out, d_qkv, scale);
        CN_CHECK_LAUNCH();
        return CN_OK;
    }
    const size_t per_wave = (size_t)(4 * CAP * 68 + (CAP <= 32 ? 4 : 2) * CAP * CAP) * sizeof(float);
    int wpb = (int)(65536 / per_wave); wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
    int per_cu = (int)((160 * 1024) / (per_wave * wpb)); per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    int blocks = (B * 8 + wpb - 1) / wpb;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu; // resident-sized grid walking the class list
    if (per_wave * wpb > 65536) {
        static CnLdsOptIn opt_in; // per device
        int opt_dev;
        if (opt_in.needed(&opt_dev)) {
            CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hh_attention_bwd_kernel<CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            opt_in.done(opt_dev);
        }
    }
    hipLaunchKernelGGL(hh_attention_bwd_kernel<CAP>, dim3(blocks), dim3(64 * wpb), per_wave * wpb, st, B, qkv, row_off, cls + c, cls + 4 + (size_t)c * B,
                       d_out, d_qkv, scale);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// Robot-human attention (EdgeAttention_M.att_func, selfAttn_srnn_temp_node.py:145-177) on the compacted rows: one
// wavefront per env.  The reference scores are t . s_j with t = temporal_edge_layer(robot) [64] and s_j =
// spatial_edge_layer(o_j) = Ws o_j + bs [64].  Since t . (Ws o_j + bs) = (Ws^T t) . o_j + t . bs and the softmax is
// invariant to the per-env constant t . bs, the kernel takes u = Ws^T t [256] and scores u . o_j directly: the
// [rows,256]x[256,64] projection of every human row (and its two backward products) is replaced by a [E,64]x[64,256]
// product per env.  masked_fill(-1e9) + softmax gives padded humans exactly zero weight (exp underflows to 0), so the
// softmax and the weighted sum run over the nd live rows only.
// (both kernels: the rows of a sample are requested eight at a time ahead of the wave-wide reductions that consume them -- as a loop of
// "load row j, reduce" every row paid its own memory round trip -- and the first eight stay in registers for the second pass: most samples
// have no more.  Same operations in the same order as the plain loops: bit-identical results.)
struct HrRows { float v[8][4]; };
__device__ __forceinline__ void hr_load8(HrRows &r, const float *__restrict__ out_sp, int r0, int j0, int nd, int lane)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float *row = out_sp + (size_t)(r0 + max(min(j0 + q, nd - 1), 0)) * 256 + lane; // (rows past nd repeat the last one: loaded, never used)
#pragma unroll
        for (int c = 0; c < 4; ++c) r.v[q][c] = row[64 * c];
    }
}
__global__ __launch_bounds__(256) void hr_attention_kernel(int E, int H, const float *__restrict__ u, int u_ld, const float *__restrict__ out_sp,
                                                           const int *__restrict__ row_off, float *__restrict__ hr_out, float *__restrict__ hr_attn)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 4 + wave;
    if (e >= E) return;
    const int r0 = row_off[e], nd = row_off[e + 1] - r0;
    const float *ue = u + (size_t)e * u_ld;
    const float u0 = ue[lane], u1 = ue[64 + lane], u2 = ue[128 + lane], u3 = ue[192 + lane];
    const float temp = (float)H / 8.0f; // temperature = num_edges / sqrt(attention_size = 64)
    float s = -INFINITY; // lane j holds the score of human j
    HrRows k0;
    hr_load8(k0, out_sp, r0, 0, nd, lane);
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < nd) {
            const float tot = wv_sum(u0 * k0.v[q][0] + u1 * k0.v[q][1] + u2 * k0.v[q][2] + u3 * k0.v[q][3]);
            if (lane == q) s = tot * temp;
        }
    for (int j0 = 8; j0 < nd; j0 += 8) {
        HrRows r;
        hr_load8(r, out_sp, r0, j0, nd, lane);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + q < nd) {
                const float tot = wv_sum(u0 * r.v[q][0] + u1 * r.v[q][1] + u2 * r.v[q][2] + u3 * r.v[q][3]);
                if (lane == j0 + q) s = tot * temp;
            }
    }
    const float mx = wv_max(s);
    const float p = lane < nd ? expf(s - mx) : 0.0f;
    const float denom = wv_sum(p);
    const float a = p / denom;
    if (hr_attn && lane < H) hr_attn[(size_t)e * H + lane] = a;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < nd) {
            const float aj = wv_readlane(a, q);
            o0 += aj * k0.v[q][0]; o1 += aj * k0.v[q][1]; o2 += aj * k0.v[q][2]; o3 += aj * k0.v[q][3];
        }
    for (int j0 = 8; j0 < nd; j0 += 8) {
        HrRows r;
        hr_load8(r, out_sp, r0, j0, nd, lane);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + q < nd) {
                const float aj = wv_readlane(a, j0 + q);
                o0 += aj * r.v[q][0]; o1 += aj * r.v[q][1]; o2 += aj * r.v[q][2]; o3 += aj * r.v[q][3];
            }
    }
    float *o = hr_out + (size_t)e * 256;
    o[lane] = o0; o[64 + lane] = o1; o[128 + lane] = o2; o[192 + lane] = o3;
}

// Backward of hr_attention_kernel for the PPO update: one wavefront per sample on the compacted rows.
//   a_j = T (u . o_j), p = softmax(a), hr = sum_j p_j o_j         (T = H / 8)
//   dp_j = d_hr . o_j ; g_j = T p_j (dp_j - sum_k p_k dp_k) ; d_u = sum_j g_j o_j ; d_o_j = p_j d_hr + g_j u
__global__ __launch_bounds__(256) void hr_attention_bwd_kernel(int B, int H, const float *__restrict__ u, const float *__restrict__ out_sp,
                                                               const int *__restrict__ row_off, const float *__restrict__ attn,
                                                               const float *__restrict__ d_hr, float *__restrict__ d_u, float *__restrict__ d_o, int u_ld, int du_ld)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 4 + wave;
    if (e >= B) return;
    const int r0 = row_off[e], nd = row_off[e + 1] - r0;
    const float a = lane < nd ? attn[(size_t)e * H + lane] : 0.0f; // lanes = humans
    const float *g = d_hr + (size_t)e * 256, *ue = u + (size_t)e * u_ld;
    const float g0 = g[lane], g1 = g[64 + lane], g2 = g[128 + lane], g3 = g[192 + lane];
    const float u0 = ue[lane], u1 = ue[64 + lane], u2 = ue[128 + lane], u3 = ue[192 + lane];
    float dp = 0.0f;
    HrRows k0;
    hr_load8(k0, out_sp, r0, 0, nd, lane);
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < nd) {
            const float tot = wv_sum(g0 * k0.v[q][0] + g1 * k0.v[q][1] + g2 * k0.v[q][2] + g3 * k0.v[q][3]);
            if (lane == q) dp = tot;
        }
    for (int j0 = 8; j0 < nd; j0 += 8) {
        HrRows r;
        hr_load8(r, out_sp, r0, j0, nd, lane);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + q < nd) {
                const float tot = wv_sum(g0 * r.v[q][0] + g1 * r.v[q][1] + g2 * r.v[q][2] + g3 * r.v[q][3]);
                if (lane == j0 + q) dp = tot;
            }
    }
    const float dot = wv_sum(a * dp);
    const float gg = a * (dp - dot) * ((float)H / 8.0f);
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    auto second = [&](const HrRows &r, int j0) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + q < nd) {
                const float aj = wv_readlane(a, j0 + q), gj = wv_readlane(gg, j0 + q);
                float *dor = d_o + (size_t)(r0 + j0 + q) * 256;
                d0 += gj * r.v[q][0]; d1 += gj * r.v[q][1]; d2 += gj * r.v[q][2]; d3 += gj * r.v[q][3];
                dor[lane] = aj * g0 + gj * u0; dor[64 + lane] = aj * g1 + gj * u1; dor[128 + lane] = aj * g2 + gj * u2; dor[192 + lane] = aj * g3 + gj * u3;
            }
    };
    second(k0, 0);
    for (int j0 = 8; j0 < nd; j0 += 8) {
        HrRows r;
        hr_load8(r, out_sp, r0, j0, nd, lane);
        second(r, j0);
    }
    float *du = d_u + (size_t)e * du_ld;
    du[lane] = d0; du[64 + lane] = d1; du[128 + lane] = d2; du[192 + lane] = d3;
}

// test tap: scatter the compacted [rows,256] activations back to [E,H,256] (zeros on padded humans)
// args.sort_humans = False: the visible humans of a sample moved to the front (stable), the others behind them; detected = max(1, visible)
// (an all-invisible sample keeps human 0: selfAttn_srnn_temp_node.py:381-383).  One wavefront per sample, lane = human.
__global__ __launch_bounds__(64) void compact_visible_kernel(int B, int H, int D, const float *__restrict__ se, const uint8_t *__restrict__ vis,
                                                             float *__restrict__ out, float *__restrict__ det)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const bool isH = lane < H;
    unsigned long long m = __ballot(isH && vis[(size_t)b * H + (isH ? lane : 0)] != 0);
    if (m == 0ull) m = 1ull;
    const unsigned long long valid = H >= 64 ? ~0ull : ((1ull << H) - 1ull);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int cnt = __popcll(m);
    const bool v = (m >> lane) & 1ull;
    const int rank = v ? __popcll(m & below) : cnt + __popcll(~m & valid & below);
    if (isH) {
        const float *src = se + ((size_t)b * H + lane) * D;
        float *dst = out + ((size_t)b * H + rank) * D;
        for (int d = 0; d < D; ++d) dst[d] = src[d];
    }
    if (lane == 0) det[b] = (float)cnt;
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(int E, int H, const float *__restrict__ src, const int *__restrict__ row_off,
                                                           float *__restrict__ dst)
{
    const int e = blockIdx.x, c = threadIdx.x;
    const int r0 = row_off[e], nd = row_off[e + 1] - r0;
    for (int j = 0; j < H; ++j) dst[((size_t)e * H + j) * 256 + c] = j < nd ? src[(size_t)(r0 + j) * 256 + c] : 0.0f;
}

// GRU cell pointwise part (PyTorch formulation, gate order r,z,n) with the done mask applied to h
// (rl/networks/srnn_model.py:43-46): gi = x W_ih^T + b_ih (bias already added), gh_raw = h W_hh^T (no bias, unmasked).
__global__ __launch_bounds__(128) void gru_pointwise_kernel(int E, const float *__restrict__ gi, const float *__restrict__ gh_raw,
                                                            const float *__restrict__ b_hh, const float *__restrict__ h_in,
                                                            const float *__restrict__ masks, float *__restrict__ h_out)
{
    const int e = blockIdx.x, c = threadIdx.x;
    if (e >= E) return;
    const float m = masks[e];
    const float *gie = gi + (size_t)e * 384, *ghe = gh_raw + (size_t)e * 384;
    const float hr = m * ghe[c] + b_hh[c], hz = m * ghe[128 + c] + b_hh[128 + c], hn = m * ghe[256 + c] + b_hh[256 + c];
    const float r = 1.0f / (1.0f + expf(-(gie[c] + hr)));
    const float z = 1.0f / (1.0f + expf(-(gie[128 + c] + hz)));
    const float n = tanhf(gie[256 + c] + r * hn);
    const float h = m * h_in[(size_t)e * 128 + c];
    h_out[(size_t)e * 128 + c] = (1.0f - z) * n + z * h;
}

// critic_linear + DiagGaussian head (model.py:64-72, distributions.py:36-44,76-95): one wavefront per env.
// ac [E,512]: columns 0..255 actor features, 256..511 critic features.
__global__ __launch_bounds__(256) void gauss_head_kernel(int E, const float *__restrict__ ac, int ld, const float *__restrict__ wv,
                                                         const float *__restrict__ bv, const float *__restrict__ wm,
                                                         const float *__restrict__ bm, const float *__restrict__ logstd,
                                                         const float *__restrict__ eps, float *__restrict__ value,
                                                         float *__restrict__ action, float *__restrict__ logp)
{
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float *a = ac + (size_t)e * ld, *c = a + 256;
    float sv = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = lane + 64 * k;
        sv += c[d] * wv[d];
        s0 += a[d] * wm[d];
        s1 += a[d] * wm[256 + d];
    }
    sv = wv_sum(sv); s0 = wv_sum(s0); s1 = wv_sum(s1);
    if (lane == 0) {
        value[e] = sv + bv[0];
        if (action) {
            const float mean0 = s0 + bm[0], mean1 = s1 + bm[1];
            const float ls0 = logstd[0], ls1 = logstd[1];
            const float sd0 = expf(ls0), sd1 = expf(ls1);
            const float a0 = eps ? mean0 + sd0 * eps[2 * e] : mean0;
            const float a1 = eps ? mean1 + sd1 * eps[2 * e + 1] : mean1;
            action[2 * e] = a0; action[2 * e + 1] = a1;
            const float HALF_LOG_2PI = 0.91893853320467274178f;
            const float d0 = a0 - mean0, d1 = a1 - mean1;
            logp[e] = (-(d0 * d0) / (2.0f * sd0 * sd0) - ls0 - HALF_LOG_2PI) + (-(d1 * d1) / (2.0f * sd1 * sd1) - ls1 - HALF_LOG_2PI);
        }
    }
}

// Weight folding: C[n][k] = scale * sum_j A[n][j] * B[j][k]  (fp64 accumulation), A [N,J], B [J,K]
__global__ void fold_mm_kernel(int N, int J, int K, const float *__restrict__ A, const float *__restrict__ B, float scale, float *__restrict__ C)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (k >= K || n >= N) return;
    double acc = 0.0;
    for (int j = 0; j < J; ++j) acc += (double)A[(size_t)n * J + j] * (double)B[(size_t)j * K + k];
    C[(size_t)n * K + k] = (float)(acc * (double)scale);
}
// c[n] = scale * (sum_j A[n][j] * b[j] + b2[n])
__global__ void fold_bias_kernel(int N, int J, const float *__restrict__ A, const float *__restrict__ b, const float *__restrict__ b2,
                                 float scale, float *__restrict__ c)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double acc = b2[n];
    for (int j = 0; j < J; ++j) acc += (double)A[(size_t)n * J + j] * (double)b[j];
    c[n] = (float)(acc * (double)scale);
}

// C[N,K] = A^T B with A [J,N], B [J,K];  c[n] = sum_j A[j][n] * b[j]
__global__ void fold_mm_tn_kernel(int N, int J, int K, const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (k >= K || n >= N) return;
    double acc = 0.0;
    for (int j = 0; j < J; ++j) acc += (double)A[(size_t)j * N + n] * (double)B[(size_t)j * K + k];
    C[(size_t)n * K + k] = (float)acc;
}
__global__ void fold_bias_tn_kernel(int N, int J, const float *__restrict__ A, const float *__restrict__ b, float *__restrict__ c)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double acc = 0.0;
    for (int j = 0; j < J; ++j) acc += (double)A[(size_t)j * N + n] * (double)b[j];
    c[n] = (float)acc;
}

constexpr size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

} // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// Training path of the robot-node sequence (cn_rn_seq_fwd / cn_rn_seq_bwd below): small helpers
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {

// critic_linear + DiagGaussian.log_probs of GIVEN actions (model.py:82-90, distributions.py:36-44): one wavefront per sample
__global__ __launch_bounds__(256) void rn_head_fwd_kernel(int B, const float *__restrict__ ac, const float *__restrict__ wv, const float *__restrict__ bv,
                                                          const float *__restrict__ wm, const float *__restrict__ bm, const float *__restrict__ logstd,
                                                          const float *__restrict__ actions, float *__restrict__ value, float *__restrict__ logp)
{
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= B) return;
    const float *a = ac + (size_t)e * 512, *c = a + 256;
    float sv = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = lane + 64 * k;
        sv += c[d] * wv[d]; s0 += a[d] * wm[d]; s1 += a[d] * wm[256 + d];
    }
    sv = wv_sum(sv); s0 = wv_sum(s0); s1 = wv_sum(s1);
    if (lane == 0) {
        value[e] = sv + bv[0];
        const float mean0 = s0 + bm[0], mean1 = s1 + bm[1], ls0 = logstd[0], ls1 = logstd[1];
        const float sd0 = expf(ls0), sd1 = expf(ls1);
        const float HALF_LOG_2PI = 0.91893853320467274178f;
        const float d0 = actions[2 * e] - mean0, d1 = actions[2 * e + 1] - mean1;
        logp[e] = (-(d0 * d0) / (2.0f * sd0 * sd0) - ls0 - HALF_LOG_2PI) + (-(d1 * d1) / (2.0f * sd1 * sd1) - ls1 - HALF_LOG_2PI);
    }
}

// Backward of the heads AND of the second trunk layers' tanh: from d_value [B], d_logp [B]
//   d_mean_j = d_logp (a_j - mean_j) / sd_j^2 ; d_logstd_j += d_logp ((a_j - mean_j)^2 / sd_j^2 - 1)
//   d2[:, 0:256]   = (d_mean_0 wm[0] + d_mean_1 wm[1]) (1 - actor^2) ;  d2[:, 256:512] = d_value wv (1 - critic^2)
// and the heads' own weight gradients (they are reductions over all B samples into 3 x 256 + 5 numbers): every workgroup keeps its sums in
// registers and writes ONE partial row; rn_reduce_rows_kernel adds the rows in order (deterministic).
constexpr int RN_HEAD_COLS = 3 * 256 + 8; // d fm_w[0] | d fm_w[1] | d cl_w | d fm_b (2) d cl_b d logstd (2) pad (3)
__global__ __launch_bounds__(256) void rn_head_bwd_kernel(int B, const float *__restrict__ ac, const float *__restrict__ wv, const float *__restrict__ wm,
                                                          const float *__restrict__ bm, const float *__restrict__ logstd, const float *__restrict__ actions,
                                                          const float *__restrict__ d_value, const float *__restrict__ d_logp, float *__restrict__ d2,
                                                          float *__restrict__ partials)
{
    __shared__ float red[4][RN_HEAD_COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w0[4], w1[4], wc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int d = lane + 64 * k; w0[k] = wm[d]; w1[k] = wm[256 + d]; wc[k] = wv[d]; }
    const float ls0 = logstd[0], ls1 = logstd[1];
    const float iv0 = expf(-2.0f * ls0), iv1 = expf(-2.0f * ls1);
    float g0[4] = {0.f, 0.f, 0.f, 0.f}, g1[4] = {0.f, 0.f, 0.f, 0.f}, gc[4] = {0.f, 0.f, 0.f, 0.f};
    float sb0 = 0.f, sb1 = 0.f, sbc = 0.f, sl0 = 0.f, sl1 = 0.f;
    for (int e = blockIdx.x * 4 + wave; e < B; e += gridDim.x * 4) {
        const float *a = ac + (size_t)e * 512, *c = a + 256;
        float av[4], cv[4], s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int d = lane + 64 * k; av[k] = a[d]; cv[k] = c[d]; s0 += av[k] * w0[k]; s1 += av[k] * w1[k]; }
        s0 = wv_sum(s0); s1 = wv_sum(s1);
        const float dv = d_value[e], dl = d_logp[e];
        const float e0 = actions[2 * e] - (s0 + bm[0]), e1 = actions[2 * e + 1] - (s1 + bm[1]);
        const float dm0 = dl * e0 * iv0, dm1 = dl * e1 * iv1;
        float *o = d2 + (size_t)e * 512;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = lane + 64 * k;
            o[d] = (dm0 * w0[k] + dm1 * w1[k]) * (1.0f - av[k] * av[k]);
            o[256 + d] = dv * wc[k] * (1.0f - cv[k] * cv[k]);
            g0[k] += dm0 * av[k]; g1[k] += dm1 * av[k]; gc[k] += dv * cv[k];
        }
        sb0 += dm0; sb1 += dm1; sbc += dv; sl0 += dl * (e0 * e0 * iv0 - 1.0f); sl1 += dl * (e1 * e1 * iv1 - 1.0f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int d = lane + 64 * k; red[wave][d] = g0[k]; red[wave][256 + d] = g1[k]; red[wave][512 + d] = gc[k]; }
    if (lane == 0) { red[wave][768] = sb0; red[wave][769] = sb1; red[wave][770] = sbc; red[wave][771] = sl0; red[wave][772] = sl1; red[wave][773] = red[wave][774] = red[wave][775] = 0.f; }
    __syncthreads();
    for (int j = threadIdx.x; j < RN_HEAD_COLS; j += 256)
        partials[(size_t)blockIdx.x * RN_HEAD_COLS + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
}

// out[j] = sum over the rows of part[R][Cn], in a fixed order: a workgroup owns 16 columns, its sixteen thread groups each walk every
// sixteenth row (independent loads) and meet in LDS as a fixed binary tree.  rl_layout: the columns are robot_linear's [10][256] partials
// (9 weights + bias, feature innermost) and land in dW [256,9] / db [256] (out = dW, out2 = db).
__global__ __launch_bounds__(256) void rn_reduce_rows_kernel(int R, int Cn, const float *__restrict__ part, float *__restrict__ out, float *__restrict__ out2,
                                                             int rl_layout)
{
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4, j = blockIdx.x * 16 + c;
    float s = 0.f;
    if (j < Cn)
        for (int r = g; r < R; r += 16) s += part[(size_t)r * Cn + j];
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && j < Cn) {
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = red[k][c];
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int k = 0; k < w; ++k) t[k] = t[k] + t[k + w];
        if (!rl_layout) out[j] = t[0];
        else { const int q = j >> 8, n = j & 255; if (q < 9) out[n * 9 + q] = t[0]; else out2[n] = t[0]; }
    }
}

// robot_linear.0's weight gradient: dW [256,9] and db [256] from drs [B,256] (already gated by the ReLU) and the 9 inputs
// (temporal_edges 2 | robot_node 7).  thread = output feature; every workgroup writes one partial [10][256] (9 weights + bias, feature innermost)
__global__ __launch_bounds__(256) void rn_rl_wgrad_kernel(int B, const float *__restrict__ drs, const float *__restrict__ temporal,
                                                          const float *__restrict__ robot_node, float *__restrict__ partials)
{
    const int n = threadIdx.x;
    float acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int e = blockIdx.x; e < B; e += gridDim.x) {
        const float d = drs[(size_t)e * 256 + n];
        acc[0] += d * temporal[e * 2]; acc[1] += d * temporal[e * 2 + 1];
#pragma unroll
        for (int q = 0; q < 7; ++q) acc[2 + q] += d * robot_node[e * 7 + q];
        acc[9] += d;
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) partials[((size_t)blockIdx.x * 10 + q) * 256 + n] = acc[q];
}
} // namespace

struct cn_policy {
    int H, D, maxE;
    bool weights_set;
    char *blob;
    // weight snapshot (device)
    float *emb0_w, *emb0_b, *emb2_w, *emb2_b;
    float *qkv_w, *qkv_b;   // folded [1536,512], [1536]
    float *os_w, *os_b;     // folded out_proj∘spatial_linear [256,512], [256]
    float *as_w, *as_b;     // attn.spatial_edge_layer [64,256]
    float *at_w, *at_b;     // attn.temporal_edge_layer [64,256]
    float *rl_w, *rl_b;     // robot_linear [256,9]
    float *enc_w, *enc_b, *edge_w, *edge_b; // [64,256] each
    float *wih, *whh, *bih, *bhh;           // GRU
    float *out_w, *out_b;                   // [256,128]
    float *ac0_w, *ac0_b;                   // concat(actor.0, critic.0) [512,256]
    float *a2_w, *a2_b, *c2_w, *c2_b;       // [256,256]
    float *cl_w, *cl_b, *fm_w, *fm_b, *logstd;
    float *te_w, *te_b;       // [spatial_edge_layer^T * temporal_edge_layer (u, 256 rows) ; encoder_linear (64 rows)] stacked [320,256]
    float *ac0f_w, *ac0f_b;   // (actor.0 ; critic.0) folded with output_linear [512,128]
    float *z;                 // [E,192] = [t_emb | relu(enc) | relu(edge)]
    // activations
    float *emb1, *emb2, *qkv, *attn, *out_sp;
    __bf16 *emb2_hi, *emb2_lo, *qkv_hi, *qkv_lo, *os_hi, *os_lo; // split copies of the three big weight matrices
    float *r_te, *r_whh, *r_edge, *r_wih, *r_ac0, *r_a2, *r_c2; // MFMA-fragment images of the robot-node weights (rn_fused.hip)
    bool self_attn = true;       // args.use_self_attn: false = spatial_linear is a two-layer MLP on the spatial edges (cn_policy_set_self_attention)
    bool taps_on;                // fused mode: write the test taps (robot_emb, hr_attn, hr_out, actor_feat) of every forward
    void *f_emb2, *f_qkv, *f_os; // MFMA-fragment images of the three big weight matrices for the fused human-human kernel
    int gemm_mode; // 0 = exact fp32 MFMA, 1 = bf16x3 split as separate launches, 2 = bf16x3 split, fused human-human kernel (default)
    unsigned long long *live_total; // device counter: sum of live rows over the profiled forwards
    int *row_off; // [maxE + 1]
    int *cls_cnt, *cls_list; // big attention size classes: [2] counts, [2][E] env lists exclusive prefix of live humans per env; row_off[E] = live rows
    float *robot_states, *t_emb, *hr_out, *hr_attn, *x, *gi, *gh, *hnew, *rnn_out, *ac1, *ac2;
    // robot-side launches that do not depend on the human-human block run on this stream, beside the big GEMMs
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    // profiling of the dominant kernel: every prof_every-th forward is bracketed by a pair of events (an event record on the
    // critical stream costs a few microseconds of dispatch gap, so the bracket is sampled, not put around every launch)
    bool profiling;       // THIS forward is bracketed
    int prof_every;       // 0 = off
    long long prof_tick;
    static constexpr int PROF_RING = 64;
    hipEvent_t ev[PROF_RING][2]; // ring of (start, stop) pairs around the QKV projection launch
    int ev_head, ev_tail;        // [tail, head) are recorded but not yet harvested
    double prof_ms[8];
    int64_t prof_n[8];
    std::vector<float> prof_samples; // the harvested brackets one by one [ms], in launch order (cn_policy_get_profile_samples)
    // called right behind the launch of the human-human kernel (cn_policy_set_post_hh_hook): side work that must not reach the CUs before it
    int (*post_hh_hook)(void *arg, void *stream);
    void *post_hh_arg;
};

// big-M GEMMs (rows = live humans): 128-row tiles
template <int BN, int ACT>
static int launch_gemm(int M, int N, int K, const float *A, int lda, const float *W, const float *bias, float *C, int ldc, hipStream_t st,
                       const int *m_dev = nullptr)
{
    return launch_gemm_t<128, BN, ACT>(M, N, K, A, lda, W, bias, C, ldc, st, m_dev, 1, GemmBatch{0, 0, 0, 0}, 1 << 30);
}
// per-env GEMMs (rows = envs, a few thousand): 64 x 64 tiles so that the launch still fills the 256 CUs
template <int ACT>
static int launch_gemm_env(int M, int N, int K, const float *A, int lda, const float *W, const float *bias, float *C, int ldc, hipStream_t st,
                           int nbatch = 1, GemmBatch gb = GemmBatch{0, 0, 0, 0}, int relu_from = 1 << 30)
{
    return launch_gemm_t<64, 64, ACT>(M, N, K, A, lda, W, bias, C, ldc, st, nullptr, nbatch, gb, relu_from);
}

extern "C" int cn_policy_create(int human_num, int edge_width, int max_envs, cn_policy **out)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(out, "cn_policy_create: null out");
    CN_REQUIRE(human_num >= 1 && human_num <= CN_MAX_HUMANS, "cn_policy_create: human_num must be in [1,%d]", CN_MAX_HUMANS);
    CN_REQUIRE(edge_width >= 1 && edge_width <= 16, "cn_policy_create: edge_width must be in [1,16]");
    CN_REQUIRE(max_envs >= 1, "cn_policy_create: max_envs must be positive");
    cn_policy *p = new (std::nothrow) cn_policy{};
    CN_REQUIRE(p, "cn_policy_create: out of host memory");
    p->H = human_num; p->D = edge_width; p->maxE = max_envs;
    const size_t E = max_envs, M = E * human_num, D = edge_width;
    size_t off = 0;
    auto carve = [&](size_t nfloat) { size_t o = off; off += align_up(nfloat * sizeof(float)); return o; };
    const size_t o_emb0w = carve(128 * D), o_emb0b = carve(128), o_emb2w = carve(512 * 128), o_emb2b = carve(512);
    const size_t o_qkvw = carve(1536 * 512), o_qkvb = carve(1536), o_osw = carve(256 * 512), o_osb = carve(256);
    const size_t o_asw = carve(64 * 256), o_asb = carve(64), o_atw = carve(64 * 256), o_atb = carve(64);
    const size_t o_rlw = carve(256 * 9), o_rlb = carve(256);
    const size_t o_encw = carve(64 * 256), o_encb = carve(64), o_edgew = carve(64 * 256), o_edgeb = carve(64);
    const size_t o_wih = carve(384 * 128), o_whh = carve(384 * 128), o_bih = carve(384), o_bhh = carve(384);
    const size_t o_outw = carve(256 * 128), o_outb = carve(256);
    const size_t o_ac0w = carve(512 * 256), o_ac0b = carve(512);
    const size_t o_a2w = carve(256 * 256), o_a2b = carve(256), o_c2w = carve(256 * 256), o_c2b = carve(256);
    const size_t o_clw = carve(256), o_clb = carve(1), o_fmw = carve(512), o_fmb = carve(2), o_ls = carve(2);
    const size_t o_emb1 = carve(M * 128), o_emb2 = carve(M * 512), o_qkv = carve(M * 1536), o_attn = carve(M * 512);
    const size_t o_outsp = carve(M * 256);
    const size_t o_rs = carve(E * 256), o_temb = carve(E * 64), o_hr = carve(E * 256), o_hra = carve(M), o_x = carve(E * 128);
    const size_t o_gi = carve(E * 384), o_gh = carve(E * 384), o_hn = carve(E * 128), o_ro = carve(E * 256);
    const size_t o_ac1 = carve(E * 512), o_ac2 = carve(E * 512);
    const size_t o_roff = carve(E + 1);
    const size_t o_live = carve(2);
    const size_t o_ccnt = carve(2), o_clist = carve(2 * E);
    const size_t o_tew = carve(320 * 256), o_teb = carve(320), o_acfw = carve(512 * 128), o_acfb = carve(512), o_z = carve(E * 384);
    const size_t o_e2h = carve(512 * 128 / 2), o_e2l = carve(512 * 128 / 2), o_qh = carve(1536 * 512 / 2), o_ql = carve(1536 * 512 / 2);
    const size_t o_osh = carve(256 * 512 / 2), o_osl = carve(256 * 512 / 2);
    const size_t o_rte = carve(320 * 256), o_rwhh = carve(384 * 128), o_redge = carve(64 * 256), o_rwih = carve(384 * 128), o_rac0 = carve(512 * 128);
    const size_t o_ra2 = carve(256 * 256), o_rc2 = carve(256 * 256);
    const size_t o_fe2 = carve(HH_EMB2_FRAG_BYTES / 4), o_fqkv = carve(HH_QKV_FRAG_BYTES / 4), o_fos = carve(HH_OS_FRAG_BYTES / 4);
    char *base = nullptr;
    hipError_t herr = hipMalloc((void **)&base, off);
    if (herr != hipSuccess) { delete p; cn_set_error("cn_policy_create: hipMalloc(%zu) failed: %s", off, hipGetErrorString(herr)); return CN_ERR_HIP; }
    p->blob = base;
    auto F = [&](size_t o) { return (float *)(base + o); };
    p->emb0_w = F(o_emb0w); p->emb0_b = F(o_emb0b); p->emb2_w = F(o_emb2w); p->emb2_b = F(o_emb2b);
    p->qkv_w = F(o_qkvw); p->qkv_b = F(o_qkvb); p->os_w = F(o_osw); p->os_b = F(o_osb);
    p->as_w = F(o_asw); p->as_b = F(o_asb); p->at_w = F(o_atw); p->at_b = F(o_atb);
    p->rl_w = F(o_rlw); p->rl_b = F(o_rlb); p->enc_w = F(o_encw); p->enc_b = F(o_encb); p->edge_w = F(o_edgew); p->edge_b = F(o_edgeb);
    p->wih = F(o_wih); p->whh = F(o_whh); p->bih = F(o_bih); p->bhh = F(o_bhh); p->out_w = F(o_outw); p->out_b = F(o_outb);
    p->ac0_w = F(o_ac0w); p->ac0_b = F(o_ac0b); p->a2_w = F(o_a2w); p->a2_b = F(o_a2b); p->c2_w = F(o_c2w); p->c2_b = F(o_c2b);
    p->cl_w = F(o_clw); p->cl_b = F(o_clb); p->fm_w = F(o_fmw); p->fm_b = F(o_fmb); p->logstd = F(o_ls);
    p->emb1 = F(o_emb1); p->emb2 = F(o_emb2); p->qkv = F(o_qkv); p->attn = F(o_attn); p->out_sp = F(o_outsp);
    p->robot_states = F(o_rs); p->t_emb = F(o_temb); p->hr_out = F(o_hr); p->hr_attn = F(o_hra); p->x = F(o_x);
    p->gi = F(o_gi); p->gh = F(o_gh); p->hnew = F(o_hn); p->rnn_out = F(o_ro); p->ac1 = F(o_ac1); p->ac2 = F(o_ac2);
    p->row_off = (int *)(base + o_roff);
    p->live_total = (unsigned long long *)(base + o_live);
    p->cls_cnt = (int *)(base + o_ccnt); p->cls_list = (int *)(base + o_clist);
    (void)hipMemset(p->live_total, 0, 8);
    p->emb2_hi = (__bf16 *)(base + o_e2h); p->emb2_lo = (__bf16 *)(base + o_e2l); p->qkv_hi = (__bf16 *)(base + o_qh); p->qkv_lo = (__bf16 *)(base + o_ql);
    p->os_hi = (__bf16 *)(base + o_osh); p->os_lo = (__bf16 *)(base + o_osl);
    p->r_te = F(o_rte); p->r_whh = F(o_rwhh); p->r_edge = F(o_redge); p->r_wih = F(o_rwih); p->r_ac0 = F(o_rac0); p->r_a2 = F(o_ra2); p->r_c2 = F(o_rc2);
    p->taps_on = true;
    p->f_emb2 = base + o_fe2; p->f_qkv = base + o_fqkv; p->f_os = base + o_fos;
    p->gemm_mode = 2;
    p->te_w = F(o_tew); p->te_b = F(o_teb); p->ac0f_w = F(o_acfw); p->ac0f_b = F(o_acfb); p->z = F(o_z);
    p->weights_set = false;
    p->profiling = false; p->prof_every = 0; p->prof_tick = 0;
    p->ev_head = p->ev_tail = 0;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest); // side work yields to the caller's stream (critical path)
    if (hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio_least) != hipSuccess || hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(base); delete p; cn_set_error("cn_policy_create: side stream / event creation failed"); return CN_ERR_HIP;
    }
    for (int i = 0; i < cn_policy::PROF_RING; ++i)
        if (hipEventCreate(&p->ev[i][0]) != hipSuccess || hipEventCreate(&p->ev[i][1]) != hipSuccess) {
            (void)hipFree(base); delete p; cn_set_error("cn_policy_create: hipEventCreate failed"); return CN_ERR_HIP;
        }
    *out = p;
    return CN_OK;
}

extern "C" int cn_policy_destroy(cn_policy *p)
{
    if (!p) return CN_OK;
    for (int i = 0; i < cn_policy::PROF_RING; ++i) { (void)hipEventDestroy(p->ev[i][0]); (void)hipEventDestroy(p->ev[i][1]); }
    (void)hipEventDestroy(p->ev_fork); (void)hipEventDestroy(p->ev_join); (void)hipStreamDestroy(p->side);
    if (p->blob) CN_HIP(hipFree(p->blob));
    delete p;
    return CN_OK;
}

#define CN_D2D(dst, src, n) CN_HIP(hipMemcpyAsync((dst), (src), (size_t)(n) * sizeof(float), hipMemcpyDeviceToDevice, st))

extern "C" int cn_policy_set_weights(cn_policy *p, const cn_policy_weights *w, void *stream)
{
    CN_REQUIRE(p && w, "cn_policy_set_weights: null argument");
    const float *const *ptrs = reinterpret_cast<const float *const *>(w);
    const size_t attn_first = offsetof(cn_policy_weights, emb2_w) / sizeof(const float *), attn_last = offsetof(cn_policy_weights, out_proj_b) / sizeof(const float *);
    for (size_t i = 0; i < sizeof(cn_policy_weights) / sizeof(const float *); ++i) {
        if (!p->self_attn && i >= attn_first && i <= attn_last) continue; // no human-human attention: those layers do not exist
        CN_REQUIRE(ptrs[i] != nullptr, "cn_policy_set_weights: weight pointer #%zu is null", i);
    }
    hipStream_t st = (hipStream_t)stream;
    const int D = p->D;
    CN_D2D(p->emb0_w, w->emb0_w, 128 * D); CN_D2D(p->emb0_b, w->emb0_b, 128);
    if (!p->self_attn) {
        // use_self_attn = False: out_sp = relu(W2 relu(W0 x + b0) + b2); W2 = spatial_linear.2 [256,128] takes the place of the folded
        // out_proj o spatial_linear image (fp32 + bf16 hi / lo planes)
        CN_D2D(p->os_w, w->spatial_linear_w, 256 * 128); CN_D2D(p->os_b, w->spatial_linear_b, 256);
        const size_t n = 256 * 128;
        hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, p->os_w, p->os_hi, p->os_lo);
        CN_CHECK_LAUNCH();
    } else {
    CN_D2D(p->emb2_w, w->emb2_w, 512 * 128); CN_D2D(p->emb2_b, w->emb2_b, 512);
    // fold (q|k|v)_linear into in_proj:  y = W_in (W_x e + b_x) + b_in ; q additionally scaled by 1/sqrt(head_dim) = 0.125
    const float *xw[3] = {w->q_w, w->k_w, w->v_w}, *xb[3] = {w->q_b, w->k_b, w->v_b};
    for (int s = 0; s < 3; ++s) {
        const float scale = s == 0 ? 0.125f : 1.0f;
        hipLaunchKernelGGL(fold_mm_kernel, dim3(2, 512), dim3(256), 0, st, 512, 512, 512, w->in_proj_w + (size_t)s * 512 * 512, xw[s], scale,
                           p->qkv_w + (size_t)s * 512 * 512);
        CN_CHECK_LAUNCH();
        hipLaunchKernelGGL(fold_bias_kernel, dim3(2), dim3(256), 0, st, 512, 512, w->in_proj_w + (size_t)s * 512 * 512, xb[s],
                           w->in_proj_b + s * 512, scale, p->qkv_b + s * 512);
        CN_CHECK_LAUNCH();
    }
    // fold out_proj into spatial_linear: y = W_sl (W_o a + b_o) + b_sl
    hipLaunchKernelGGL(fold_mm_kernel, dim3(2, 256), dim3(256), 0, st, 256, 512, 512, w->spatial_linear_w, w->out_proj_w, 1.0f, p->os_w);
    CN_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_bias_kernel, dim3(1), dim3(256), 0, st, 256, 512, w->spatial_linear_w, w->out_proj_b, w->spatial_linear_b, 1.0f, p->os_b);
    CN_CHECK_LAUNCH();
    {
        struct { const float *w; __bf16 *hi, *lo; size_t n; } sp[3] = {{p->emb2_w, p->emb2_hi, p->emb2_lo, 512 * 128},
                                                                       {p->qkv_w, p->qkv_hi, p->qkv_lo, 1536 * 512},
                                                                       {p->os_w, p->os_hi, p->os_lo, 256 * 512}};
        for (auto &x : sp) {
            hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((x.n + 255) / 256)), dim3(256), 0, st, x.n, x.w, x.hi, x.lo);
            CN_CHECK_LAUNCH();
        }
    }
    if (int rc = hh_fused_bake(p->emb2_w, p->qkv_w, p->os_w, p->f_emb2, p->f_qkv, p->f_os, st)) return rc;
    } // self_attn
    CN_D2D(p->as_w, w->attn_spatial_w, 64 * 256); CN_D2D(p->as_b, w->attn_spatial_b, 64);
    CN_D2D(p->at_w, w->attn_temporal_w, 64 * 256); CN_D2D(p->at_b, w->attn_temporal_b, 64);
    CN_D2D(p->rl_w, w->robot_linear_w, 256 * 9); CN_D2D(p->rl_b, w->robot_linear_b, 256);
    CN_D2D(p->enc_w, w->enc_w, 64 * 256); CN_D2D(p->enc_b, w->enc_b, 64);
    CN_D2D(p->edge_w, w->edge_embed_w, 64 * 256); CN_D2D(p->edge_b, w->edge_embed_b, 64);
    CN_D2D(p->wih, w->gru_w_ih, 384 * 128); CN_D2D(p->whh, w->gru_w_hh, 384 * 128);
    CN_D2D(p->bih, w->gru_b_ih, 384); CN_D2D(p->bhh, w->gru_b_hh, 384);
    CN_D2D(p->out_w, w->out_w, 256 * 128); CN_D2D(p->out_b, w->out_b, 256);
    CN_D2D(p->ac0_w, w->actor0_w, 256 * 256); CN_D2D(p->ac0_w + 256 * 256, w->critic0_w, 256 * 256);
    CN_D2D(p->ac0_b, w->actor0_b, 256); CN_D2D(p->ac0_b + 256, w->critic0_b, 256);
    CN_D2D(p->a2_w, w->actor2_w, 256 * 256); CN_D2D(p->a2_b, w->actor2_b, 256);
    CN_D2D(p->c2_w, w->critic2_w, 256 * 256); CN_D2D(p->c2_b, w->critic2_b, 256);
    // u = Ws^T (Wt r + bt): the robot-human scores become u . o_j (see hr_attention_kernel)
    hipLaunchKernelGGL(fold_mm_tn_kernel, dim3(1, 256), dim3(256), 0, st, 256, 64, 256, w->attn_spatial_w, w->attn_temporal_w, p->te_w);
    CN_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_bias_tn_kernel, dim3(1), dim3(256), 0, st, 256, 64, w->attn_spatial_w, w->attn_temporal_b, p->te_b);
    CN_CHECK_LAUNCH();
    CN_D2D(p->te_w + 256 * 256, w->enc_w, 64 * 256);
    CN_D2D(p->te_b + 256, w->enc_b, 64);
    // fold output_linear into the first actor / critic layers: tanh(W0 (Wo h + bo) + b0) = tanh((W0 Wo) h + (W0 bo + b0))
    hipLaunchKernelGGL(fold_mm_kernel, dim3(1, 512), dim3(128), 0, st, 512, 256, 128, p->ac0_w, w->out_w, 1.0f, p->ac0f_w);
    CN_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_bias_kernel, dim3(2), dim3(256), 0, st, 512, 256, p->ac0_w, w->out_b, p->ac0_b, 1.0f, p->ac0f_b);
    CN_CHECK_LAUNCH();
    {
        struct { int N, K; const float *w; float *out; } bk[7] = {{320, 256, p->te_w, p->r_te}, {384, 128, p->whh, p->r_whh}, {64, 256, p->edge_w, p->r_edge},
                                                                  {384, 128, p->wih, p->r_wih}, {512, 128, p->ac0f_w, p->r_ac0}, {256, 256, p->a2_w, p->r_a2},
                                                                  {256, 256, p->c2_w, p->r_c2}};
        for (auto &b : bk)
            if (int rc = rn_fused_bake(b.N, b.K, b.w, b.out, st)) return rc;
    }
    CN_D2D(p->cl_w, w->critic_linear_w, 256); CN_D2D(p->cl_b, w->critic_linear_b, 1);
    CN_D2D(p->fm_w, w->fc_mean_w, 512); CN_D2D(p->fm_b, w->fc_mean_b, 2); CN_D2D(p->logstd, w->logstd, 2);
    p->weights_set = true;
    return CN_OK;
}

// Collect finished (start, stop) pairs.  `all` waits for everything recorded; otherwise only the oldest slot is waited
// for, and only when the ring is full (it finished long ago: the host is at most a few launches ahead of the device).
static int harvest_profile(cn_policy *p, bool all)
{
    constexpr int R = cn_policy::PROF_RING;
    while (p->ev_tail != p->ev_head) {
        const bool full = ((p->ev_head + 1) % R) == p->ev_tail;
        if (!all && !full) break;
        CN_HIP(hipEventSynchronize(p->ev[p->ev_tail][1]));
        float ms = 0.f;
        CN_HIP(hipEventElapsedTime(&ms, p->ev[p->ev_tail][0], p->ev[p->ev_tail][1]));
        p->prof_ms[0] += ms; p->prof_n[0] += 1;
        if (p->prof_samples.size() < (size_t)1 << 20) p->prof_samples.push_back(ms);
        p->ev_tail = (p->ev_tail + 1) % R;
    }
    return CN_OK;
}

// use_self_attn = False (selfAttn_srnn_temp_node.py:342-345, :404-408): out_sp = relu(W2 relu(W0 x + b0) + b2) on the live rows
// (row offsets already built).  Two launches: the D -> 128 layer (embed0_kernel, exact fp32) and the 128 -> 256 layer on the large-GEMM
// kernels in the policy's arithmetic mode (bf16x3 split, or exact fp32 MFMA in mode 0).
static int spatial_mlp_forward(cn_policy *p, int E, const cn_obs *obs, hipStream_t st)
{
    const int H = p->H, D = p->D, M = E * H;
    const int *m_dev = p->row_off + E;
    const int blocks = E < 4096 ? E : 4096;
    hipLaunchKernelGGL(embed0_kernel, dim3(blocks), dim3(128), 0, st, E, H, D, obs->spatial_edges, p->emb0_w, p->emb0_b, p->row_off, p->emb1);
    CN_CHECK_LAUNCH();
    if (p->gemm_mode != 0) return launch_gemm3<128, ACT_RELU>(M, 256, 128, p->emb1, 128, p->os_hi, p->os_lo, p->os_b, p->out_sp, 256, st, m_dev);
    return launch_gemm<128, ACT_RELU>(M, 256, 128, p->emb1, 128, p->os_w, p->os_b, p->out_sp, 256, st, m_dev);
}

static int policy_forward(cn_policy *p, int E, const cn_obs *obs, const float *hxs_in, const float *masks, const float *eps,
                          float *value, float *action, float *logp, float *hxs_out, hipStream_t st)
{
    CN_REQUIRE(p, "policy: null handle");
    if (!p->weights_set) { cn_set_error("policy: call cn_policy_set_weights first"); return CN_ERR_STATE; }
    CN_REQUIRE(E >= 1 && E <= p->maxE, "policy: E=%d outside [1,%d]", E, p->maxE);
    CN_REQUIRE(obs && obs->robot_node && obs->temporal_edges && obs->spatial_edges && obs->detected_human_num, "policy: null observation pointer");
    CN_REQUIRE(hxs_in && masks && value, "policy: null pointer");
    const int H = p->H, D = p->D, M = E * H;
    int rc;
    p->profiling = p->prof_every > 0 && (p->prof_tick++ % p->prof_every) == 0;
    if (p->gemm_mode == 2) {
        // fused mode: two launches -- the human-human kernel (which also builds the row offsets) and the robot-node kernel
        if (p->profiling) { if ((rc = harvest_profile(p, false))) return rc; CN_HIP(hipEventRecord(p->ev[p->ev_head][0], st)); }
        static int hh_prio = -1;
        if (hh_prio < 0) { const char *v = getenv("CN_HH_PRIO"); hh_prio = v ? atoi(v) : 1; }
        HhFusedWeights fw{p->f_emb2, p->f_qkv, p->f_os, p->emb0_w, p->emb0_b, p->emb2_b, p->qkv_b, p->os_b, hh_prio, 1.0f, nullptr, nullptr, nullptr, nullptr};
        if (!p->self_attn) {
            hipLaunchKernelGGL(row_offsets_kernel, dim3(1), dim3(1024), 0, st, E, H, obs->detected_human_num, p->row_off,
                               p->profiling ? p->live_total : (unsigned long long *)nullptr, p->cls_cnt, p->cls_list);
            CN_CHECK_LAUNCH();
            if ((rc = spatial_mlp_forward(p, E, obs, st))) return rc;
        } else
        if ((rc = hh_fused_forward(E, H, D, obs->spatial_edges, obs->detected_human_num, p->row_off,
                                   p->profiling ? p->live_total : (unsigned long long *)nullptr, fw, p->out_sp, st, obs->row_plan))) return rc;
        if (p->profiling) { CN_HIP(hipEventRecord(p->ev[p->ev_head][1], st)); p->ev_head = (p->ev_head + 1) % cn_policy::PROF_RING; }
        if (p->post_hh_hook) { if ((rc = p->post_hh_hook(p->post_hh_arg, (void *)st))) return rc; }
        RnFusedArgs ra{};
        ra.temporal = obs->temporal_edges; ra.robot_node = obs->robot_node; ra.hxs_in = hxs_in; ra.masks = masks; ra.eps = eps;
        ra.out_sp = p->out_sp; ra.row_off = p->row_off;

        ra.rl_w = p->rl_w; ra.rl_b = p->rl_b; ra.f_te = p->r_te; ra.te_b = p->te_b; ra.f_whh = p->r_whh; ra.bhh = p->bhh;
        ra.f_edge = p->r_edge; ra.edge_b = p->edge_b; ra.f_wih = p->r_wih; ra.bih = p->bih; ra.f_ac0 = p->r_ac0; ra.ac0_b = p->ac0f_b;
        ra.f_a2 = p->r_a2; ra.a2_b = p->a2_b; ra.f_c2 = p->r_c2; ra.c2_b = p->c2_b;
        ra.cl_w = p->cl_w; ra.cl_b = p->cl_b; ra.fm_w = p->fm_w; ra.fm_b = p->fm_b; ra.logstd = p->logstd;
        ra.value = value; ra.action = action; ra.logp = logp; ra.hxs_out = hxs_out ? hxs_out : p->hnew;
        if (p->taps_on) { ra.tap_robot = p->robot_states; ra.tap_attn = p->hr_attn; ra.tap_hr = p->hr_out; ra.tap_actor = p->ac2; }
        return rn_fused_forward(E, H, ra, st);
    }
    // ---- robot node: nothing here depends on the human-human block, so it runs beside it on the side stream ----
    CN_HIP(hipEventRecord(p->ev_fork, st)); // inputs (and the previous forward's readers of z / gh) are ordered before this point
    CN_HIP(hipStreamWaitEvent(p->side, p->ev_fork, 0));
    {
        int blocks = E < 2048 ? E : 2048;
        hipLaunchKernelGGL(robot_embed_kernel, dim3(blocks), dim3(256), 0, p->side, E, obs->temporal_edges, obs->robot_node, p->rl_w, p->rl_b, p->robot_states);
        CN_CHECK_LAUNCH();
    }
    // [u | relu(enc)] in one launch (both read robot_states); z = [u (256) | enc (64) | edge (64)], GRU input x = z + 256
    if ((rc = launch_gemm_env<ACT_NONE>(E, 320, 256, p->robot_states, 256, p->te_w, p->te_b, p->z, 384, p->side, 1, GemmBatch{0, 0, 0, 0}, 256))) return rc;
    if ((rc = launch_gemm_env<ACT_NONE>(E, 384, 128, hxs_in, 128, p->whh, nullptr, p->gh, 384, p->side))) return rc; // GRU hidden-side gates
    CN_HIP(hipEventRecord(p->ev_join, p->side));
    // ---- human-human block on the compacted live rows (row_off[E] rows, known only on the device) ----
    const int *m_dev = p->row_off + E;
    // (without the human-human block no launch of this path is bracketed by a profiling event pair: the live rows are not counted either,
    // so that cn_policy_get_profile never reports rows without a matching device time)
    hipLaunchKernelGGL(row_offsets_kernel, dim3(1), dim3(1024), 0, st, E, H, obs->detected_human_num, p->row_off,
                           p->profiling && p->self_attn ? p->live_total : (unsigned long long *)nullptr, p->cls_cnt, p->cls_list);
    CN_CHECK_LAUNCH();
    if (!p->self_attn) {
        if ((rc = spatial_mlp_forward(p, E, obs, st))) return rc;
    } else {
    {
        int blocks = E < 4096 ? E : 4096;
        hipLaunchKernelGGL(embed0_kernel, dim3(blocks), dim3(128), 0, st, E, H, D, obs->spatial_edges, p->emb0_w, p->emb0_b, p->row_off, p->emb1);
        CN_CHECK_LAUNCH();
    }
    const bool split = p->gemm_mode == 1;
    if (split) rc = launch_gemm3<128, ACT_RELU>(M, 512, 128, p->emb1, 128, p->emb2_hi, p->emb2_lo, p->emb2_b, p->emb2, 512, st, m_dev);
    else rc = launch_gemm<128, ACT_RELU>(M, 512, 128, p->emb1, 128, p->emb2_w, p->emb2_b, p->emb2, 512, st, m_dev);
    if (rc) return rc;
    if (p->profiling) { if ((rc = harvest_profile(p, false))) return rc; CN_HIP(hipEventRecord(p->ev[p->ev_head][0], st)); }
    if (split) rc = launch_gemm3<128, ACT_NONE>(M, 1536, 512, p->emb2, 512, p->qkv_hi, p->qkv_lo, p->qkv_b, p->qkv, 1536, st, m_dev);
    else rc = launch_gemm<128, ACT_NONE>(M, 1536, 512, p->emb2, 512, p->qkv_w, p->qkv_b, p->qkv, 1536, st, m_dev);
    if (rc) return rc;
    if (p->profiling) { CN_HIP(hipEventRecord(p->ev[p->ev_head][1], st)); p->ev_head = (p->ev_head + 1) % cn_policy::PROF_RING; }
    {
        if ((rc = launch_hh_attention<8>(E, 0, p->qkv, p->row_off, p->attn, st))) return rc;
        if (H > 8 && (rc = launch_hh_attention<16>(E, 8, p->qkv, p->row_off, p->attn, st))) return rc;
        if (H > 16 && (rc = launch_hh_attention<32>(E, 16, p->qkv, p->row_off, p->attn, st, 1.0f, p->cls_cnt, p->cls_list))) return rc;
        if (H > 32 && (rc = launch_hh_attention<64>(E, 32, p->qkv, p->row_off, p->attn, st, 1.0f, p->cls_cnt + 1, p->cls_list + (size_t)E))) return rc;
    }
    if (split) rc = launch_gemm3<128, ACT_RELU>(M, 256, 512, p->attn, 512, p->os_hi, p->os_lo, p->os_b, p->out_sp, 256, st, m_dev);
    else rc = launch_gemm<128, ACT_RELU>(M, 256, 512, p->attn, 512, p->os_w, p->os_b, p->out_sp, 256, st, m_dev);
    if (rc) return rc;
    }
    // ---- robot-human attention (robot node embeddings arrive from the side stream) ----
    CN_HIP(hipStreamWaitEvent(st, p->ev_join, 0));
    {
        hipLaunchKernelGGL(hr_attention_kernel, dim3((E + 3) / 4), dim3(256), 0, st, E, H, p->z, 384, p->out_sp, p->row_off, p->hr_out, p->hr_attn);
        CN_CHECK_LAUNCH();
    }
    // ---- EndRNN: edge encoder -> GRU (output_linear is folded into the actor / critic trunks) ----
    if ((rc = launch_gemm_env<ACT_RELU>(E, 64, 256, p->hr_out, 256, p->edge_w, p->edge_b, p->z + 320, 384, st))) return rc;
    if ((rc = launch_gemm_env<ACT_NONE>(E, 384, 128, p->z + 256, 384, p->wih, p->bih, p->gi, 384, st))) return rc;
    float *hdst = hxs_out ? hxs_out : p->hnew;
    hipLaunchKernelGGL(gru_pointwise_kernel, dim3(E), dim3(128), 0, st, E, p->gi, p->gh, p->bhh, hxs_in, masks, hdst);
    CN_CHECK_LAUNCH();
    // ---- actor / critic trunks: first layers stacked (+ folded output_linear), second layers as one batched launch ----
    if ((rc = launch_gemm_env<ACT_TANH>(E, 512, 128, hdst, 128, p->ac0f_w, p->ac0f_b, p->ac1, 512, st))) return rc;
    if ((rc = launch_gemm_env<ACT_TANH>(E, 256, 256, p->ac1, 512, p->a2_w, p->a2_b, p->ac2, 512, st, 2,
                                        GemmBatch{256, (long long)(p->c2_w - p->a2_w), (long long)(p->c2_b - p->a2_b), 256}))) return rc;
    hipLaunchKernelGGL(gauss_head_kernel, dim3((E + 3) / 4), dim3(256), 0, st, E, p->ac2, 512, p->cl_w, p->cl_b, p->fm_w, p->fm_b, p->logstd, eps,
                       value, action, logp);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_policy_act(cn_policy *p, int E, const cn_obs *obs, const float *hxs_in, const float *masks, const float *eps,
                             float *value, float *action, float *logp, float *hxs_out, void *stream)
{
    CN_REQUIRE(action && logp && hxs_out, "cn_policy_act: null output pointer");
    CN_REQUIRE(hxs_out != hxs_in, "cn_policy_act: hxs_out must not alias hxs_in");
    return policy_forward(p, E, obs, hxs_in, masks, eps, value, action, logp, hxs_out, (hipStream_t)stream);
}

extern "C" int cn_policy_get_value(cn_policy *p, int E, const cn_obs *obs, const float *hxs_in, const float *masks, float *value, void *stream)
{
    return policy_forward(p, E, obs, hxs_in, masks, nullptr, value, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int cn_policy_get_taps(cn_policy *p, int E, float *spatial_lin, float *hr_attn, float *hr_out, float *robot_emb, float *actor_feat, void *stream)
{
    CN_REQUIRE(p && E >= 1 && E <= p->maxE, "cn_policy_get_taps: bad argument");
    if (p->gemm_mode == 2 && !p->taps_on) { cn_set_error("cn_policy_get_taps: taps are switched off (cn_policy_set_taps)"); return CN_ERR_STATE; }
    hipStream_t st = (hipStream_t)stream;
    const size_t M = (size_t)E * p->H;
    (void)M;
    if (spatial_lin) {
        hipLaunchKernelGGL(scatter_rows_kernel, dim3(E), dim3(256), 0, st, E, p->H, p->out_sp, p->row_off, spatial_lin);
        CN_CHECK_LAUNCH();
    }
    if (hr_attn) CN_D2D(hr_attn, p->hr_attn, M);
    if (hr_out) CN_D2D(hr_out, p->hr_out, (size_t)E * 256);
    if (robot_emb) CN_D2D(robot_emb, p->robot_states, (size_t)E * 256);
    if (actor_feat && p->gemm_mode == 2) CN_D2D(actor_feat, p->ac2, (size_t)E * 256); // the fused robot-node kernel taps [E,256] directly
    else if (actor_feat) CN_HIP(hipMemcpy2DAsync(actor_feat, 256 * sizeof(float), p->ac2, 512 * sizeof(float), 256 * sizeof(float), E, hipMemcpyDeviceToDevice, st));
    return CN_OK;
}

extern "C" int cn_policy_set_gemm_mode(cn_policy *p, int mode)
{
    CN_REQUIRE(p && mode >= 0 && mode <= 2, "cn_policy_set_gemm_mode: mode must be 0 (fp32 MFMA), 1 (bf16x3 split, separate launches) or 2 (bf16x3 split, fused)");
    p->gemm_mode = mode;
    return CN_OK;
}

extern "C" int cn_policy_set_self_attention(cn_policy *p, int enabled)
{
    CN_REQUIRE(p, "cn_policy_set_self_attention: null handle");
    if (p->self_attn != (enabled != 0)) { p->self_attn = enabled != 0; p->weights_set = false; }
    return CN_OK;
}

extern "C" int cn_policy_set_post_hh_hook(cn_policy *p, int (*fn)(void *arg, void *stream), void *arg)
{
    CN_REQUIRE(p, "cn_policy_set_post_hh_hook: null handle");
    p->post_hh_hook = fn; p->post_hh_arg = fn ? arg : nullptr;
    return CN_OK;
}

extern "C" int cn_policy_set_taps(cn_policy *p, int enabled)
{
    CN_REQUIRE(p, "cn_policy_set_taps: null handle");
    p->taps_on = enabled != 0;
    return CN_OK;
}

extern "C" int cn_policy_set_profiling(cn_policy *p, int enabled)
{
    CN_REQUIRE(p && enabled >= 0, "cn_policy_set_profiling: null handle or negative stride");
    p->prof_every = enabled; p->prof_tick = 0; p->profiling = false;
    return CN_OK;
}

extern "C" int cn_policy_get_profile_samples(cn_policy *p, float *ms_out, int cap)
{
    CN_REQUIRE(p && (ms_out || cap == 0) && cap >= 0, "cn_policy_get_profile_samples: null argument");
    if (int rc = harvest_profile(p, true)) return rc;
    const int n = (int)p->prof_samples.size();
    for (int i = 0; i < n && i < cap; ++i) ms_out[i] = p->prof_samples[i];
    return n;
}

extern "C" int cn_policy_reset_profile(cn_policy *p)
{
    CN_REQUIRE(p, "cn_policy_reset_profile: null handle");
    if (int rc = harvest_profile(p, true)) return rc;
    for (int i = 0; i < 8; ++i) { p->prof_ms[i] = 0.0; p->prof_n[i] = 0; }
    p->prof_samples.clear();
    p->prof_tick = 0;
    CN_HIP(hipMemset(p->live_total, 0, 8));
    return CN_OK;
}

extern "C" int cn_policy_get_profile(cn_policy *p, double *ms_out, int64_t *launches_out)
{
    CN_REQUIRE(p && ms_out && launches_out, "cn_policy_get_profile: null argument");
    if (int rc = harvest_profile(p, true)) return rc;
    unsigned long long live = 0;
    CN_HIP(hipMemcpy(&live, p->live_total, 8, hipMemcpyDeviceToHost)); // synchronising: measurement aid only
    p->prof_n[1] = (int64_t)live;                                        // [1] = live (env, human) rows summed over the profiled forwards
    for (int i = 0; i < 8; ++i) { ms_out[i] = p->prof_ms[i]; launches_out[i] = p->prof_n[i]; }
    return CN_OK;
}

// ---- the human-human block of the TRAINING forward as one launch (the rollout's fused kernel + the activations the backward needs) ----
extern "C" int64_t cn_hh_block_workspace_bytes(void) { return (int64_t)(HH_EMB2_FRAG_BYTES + HH_QKV_FRAG_BYTES + HH_OS_FRAG_BYTES); }

extern "C" int cn_hh_block_fwd(int B, int H, int D, const float *spatial_edges, const int *row_off, const float *emb0_w, const float *emb0_b,
                               const float *emb2_w, const float *emb2_b, const float *qkv_w, const float *qkv_b, const float *os_w, const float *os_b,
                               float q_scale, void *workspace, float *e0, float *x, float *qkv, float *attn, float *out_sp, void *stream)
{
    CN_REQUIRE(B >= 1 && H >= 1 && H <= 48 && D >= 1 && D <= 16, "cn_hh_block_fwd: B=%d H=%d D=%d outside B >= 1, 1 <= H <= 48, 1 <= D <= 16", B, H, D);
    CN_REQUIRE(spatial_edges && row_off && emb0_w && emb0_b && emb2_w && emb2_b && qkv_w && qkv_b && os_w && os_b && workspace && e0 && x && qkv && attn && out_sp,
               "cn_hh_block_fwd: null pointer");
    CN_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)emb2_b & 15) == 0 && ((uintptr_t)qkv_b & 15) == 0 && ((uintptr_t)os_b & 15) == 0 &&
               ((uintptr_t)e0 & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)attn & 15) == 0 && ((uintptr_t)out_sp & 15) == 0,
               "cn_hh_block_fwd: workspace, bias vectors and outputs must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    void *f_emb2 = ws, *f_qkv = ws + HH_EMB2_FRAG_BYTES, *f_os = ws + HH_EMB2_FRAG_BYTES + HH_QKV_FRAG_BYTES;
    if (int rc = hh_fused_bake(emb2_w, qkv_w, os_w, f_emb2, f_qkv, f_os, st)) return rc;
    HhFusedWeights fw{f_emb2, f_qkv, f_os, emb0_w, emb0_b, emb2_b, qkv_b, os_b, 0, q_scale, e0, x, qkv, attn};
    return hh_fused_forward(B, H, D, spatial_edges, nullptr, const_cast<int *>(row_off), nullptr, fw, out_sp, st);
}

// ---- stand-alone attention core (training path: autograd Function in the host mirror) ----
extern "C" int cn_obs_compact_visible(int B, int H, int D, const float *spatial_edges, const uint8_t *visible_masks, float *out_edges, float *out_detected,
                                      void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(B >= 1 && H >= 1 && H <= CN_MAX_HUMANS && D >= 1 && spatial_edges && visible_masks && out_edges && out_detected,
               "cn_obs_compact_visible: bad argument");
    CN_REQUIRE(out_edges != spatial_edges, "cn_obs_compact_visible: out_edges must not alias spatial_edges");
    hipLaunchKernelGGL(compact_visible_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, B, H, D, spatial_edges, visible_masks, out_edges, out_detected);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int64_t cn_hh_attention_workspace_ints(int B) { return B > 0 ? 4 + 4 * (int64_t)B : 0; }

// cls (optional, cn_hh_attention_workspace_ints(B) ints): when given, the size-class lists are built here and each class launch walks
// only its own units; cn_hh_attention_bwd can reuse the same lists (pass the buffer back, unchanged).
extern "C" int cn_hh_attention_fwd(int B, int H, const float *qkv, const int *row_off, float scale, float *out, int *cls, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(B >= 1 && H >= 1 && H <= CN_MAX_HUMANS && qkv && row_off && out, "cn_hh_attention_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (cls) {
        CN_HIP(hipMemsetAsync(cls, 0, 4 * sizeof(int), st));
        hipLaunchKernelGGL(hh_classify_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, row_off, cls);
        CN_CHECK_LAUNCH();
    }
    const int *cc = cls, *cl = cls ? cls + 4 : nullptr;
    if ((rc = launch_hh_attention<8>(B, 0, qkv, row_off, out, st, scale, cc, cl))) return rc;
    if (H > 8 && (rc = launch_hh_attention<16>(B, 8, qkv, row_off, out, st, scale, cc ? cc + 1 : nullptr, cl ? cl + (size_t)B : nullptr))) return rc;
    if (H > 16 && (rc = launch_hh_attention<32>(B, 16, qkv, row_off, out, st, scale, cc ? cc + 2 : nullptr, cl ? cl + 2 * (size_t)B : nullptr))) return rc;
    if (H > 32 && (rc = launch_hh_attention<64>(B, 32, qkv, row_off, out, st, scale, cc ? cc + 3 : nullptr, cl ? cl + 3 * (size_t)B : nullptr))) return rc;
    return CN_OK;
}

extern "C" int cn_hr_attention_fwd(int B, int H, const float *u, const float *out_sp, const int *row_off, float *hr_out, float *attn, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(B >= 1 && H >= 1 && H <= CN_MAX_HUMANS && u && out_sp && row_off && hr_out && attn, "cn_hr_attention_fwd: bad argument");
    hipLaunchKernelGGL(hr_attention_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, H, u, 256, out_sp, row_off, hr_out, attn);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_hr_attention_bwd(int B, int H, const float *u, const float *out_sp, const int *row_off, const float *attn, const float *d_hr,
                                   float *d_u, float *d_o, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(B >= 1 && H >= 1 && H <= CN_MAX_HUMANS && u && out_sp && row_off && attn && d_hr && d_u && d_o, "cn_hr_attention_bwd: bad argument");
    hipLaunchKernelGGL(hr_attention_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, H, u, out_sp, row_off, attn, d_hr, d_u, d_o, 256, 256);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

extern "C" int cn_hh_attention_bwd(int B, int H, const float *qkv, const int *row_off, const float *d_out, float scale, float *d_qkv, int *cls,
                                   int cls_ready, void *stream)
{
    if (int rc = cn_require_device()) return rc;
    CN_REQUIRE(B >= 1 && H >= 1 && H <= CN_MAX_HUMANS && qkv && row_off && d_out && d_qkv && cls, "cn_hh_attention_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (!cls_ready) { // lists not built by a preceding cn_hh_attention_fwd on the same row_off
        CN_HIP(hipMemsetAsync(cls, 0, 4 * sizeof(int), st));
        hipLaunchKernelGGL(hh_classify_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, row_off, cls);
        CN_CHECK_LAUNCH();
    }
    int rc;
    if ((rc = launch_hh_attention_bwd<8>(B, qkv, row_off, cls, 0, d_out, d_qkv, scale, st))) return rc;
    if (H > 8 && (rc = launch_hh_attention_bwd<16>(B, qkv, row_off, cls, 1, d_out, d_qkv, scale, st))) return rc;
    if (H > 16 && (rc = launch_hh_attention_bwd<32>(B, qkv, row_off, cls, 2, d_out, d_qkv, scale, st))) return rc;
    if (H > 32 && (rc = launch_hh_attention_bwd<64>(B, qkv, row_off, cls, 3, d_out, d_qkv, scale, st))) return rc;
    return CN_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// cn_rn_seq_fwd / cn_rn_seq_bwd: the robot-node sequence of evaluate_actions (see include/crowdnav_hip.h).  A sequence of the kernels of
// the separate-launch rollout forward run over all B = T * N samples at once (every layer but the GRU is independent across samples), the
// GRU as ONE launch per direction (cn_gru_seq_*), and their backward counterparts: products on the split-precision NT kernel of the big
// layers (activation or activation derivative in the epilogue), weight gradients on the split-K TN kernel (cn_linear_wgrad), small
// reductions in fixed order.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
struct RnWs { // carve-up of the backward workspace (floats)
    size_t d2, d1, dhs, dgi, dgh, dz, dhr, drs, a2T, c2T, ac0T, wihT, edgeT, teT, part, dbp, small, total;
    int small_rows;
};
RnWs rn_ws(int T, int N)
{
    const size_t B = (size_t)T * N;
    RnWs w{};
    size_t off = 0;
    auto carve = [&](size_t n) { size_t o = off; off += (n + 63) & ~size_t(63); return o; };
    w.d2 = carve(B * 512); w.d1 = carve(B * 512); w.dhs = carve(B * 128); w.dgi = carve(B * 384); w.dgh = carve(B * 384); w.dz = carve(B * 384);
    w.dhr = carve(B * 256); w.drs = carve(B * 256);
    w.a2T = carve(256 * 256); w.c2T = carve(256 * 256); w.ac0T = carve(128 * 512); w.wihT = carve(128 * 384); w.edgeT = carve(256 * 64); w.teT = carve(256 * 320);
    // weight-gradient partials: the largest of the seven products (splits <= 66 by construction of cn_linear_wgrad_splits)
    size_t pmax = 0, bmax = 0;
    const int shp[7][2] = {{256, 256}, {256, 256}, {512, 128}, {384, 128}, {384, 128}, {64, 256}, {320, 256}};
    for (auto &q : shp) {
        const size_t sp = (size_t)cn_linear_wgrad_splits((int)B, q[0], q[1]);
        pmax = pmax > sp * q[0] * q[1] ? pmax : sp * q[0] * q[1];
        bmax = bmax > sp * q[0] ? bmax : sp * q[0];
    }
    if (bmax < (size_t)RN_HEAD_COLS) bmax = RN_HEAD_COLS; // dbp also receives the reduced head gradients (cn_rn_seq_bwd): with one split per product
                                                          // (a few samples) 512 floats would let them run into `small`, which that reduction is reading
    w.part = carve(pmax); w.dbp = carve(bmax);
    w.small_rows = 1024;
    w.small = carve((size_t)w.small_rows * (RN_HEAD_COLS > 2560 ? RN_HEAD_COLS : 2560)); // head partials [1024, 776] / robot_linear partials [1024, 256, 10]
    w.total = off;
    return w;
}
int rn_wgrad(int M, int N, int K, const float *dY, int ldy, const float *X, int ldx, float *ws, const RnWs &L, float *dW, float *db, hipStream_t st)
{
    const int splits = cn_linear_wgrad_splits(M, N, K);
    CN_REQUIRE(splits >= 1, "cn_rn_seq_bwd: no split-K plan for a %d x %d weight gradient over %d rows", N, K, M);
    return cn_linear_wgrad(M, N, K, dY, ldy, nullptr, X, ldx, splits, ws + L.part, ws + L.dbp, dW, db, (void *)st);
}
template <int ACT>
int rn_gemm(int M, int N, int K, const float *A, int lda, const float *W, const float *bias, float *C, int ldc, hipStream_t st, const float *aux = nullptr, int ldaux = 0,
            int relu_from = 1 << 30)
{
    return launch_gemm_t<64, 64, ACT>(M, N, K, A, lda, W, bias, C, ldc, st, nullptr, 1, GemmBatch{0, 0, 0, 0, aux, ldaux}, relu_from);
}
// The products with N % 128 == 0 and K % 64 == 0 -- all but edge_attention_embed's forward (64 outputs) -- run on the split-precision
// kernel of the update's big layers (cn_linear_fwd_act, gemm3p.h): three bf16 MFMA products per term at ~5x the rate of the exact-fp32
// instruction.  `planes` = hi plane followed by lo plane (N * K bf16 each = the N * K floats of the slot) in that kernel's fragment order.
int rn_split(const float *w, int rows, int cols, int transpose, int n_padded, float *planes, hipStream_t st)
{
    const size_t n = (size_t)(n_padded ? n_padded : (transpose ? cols : rows)) * (transpose ? rows : cols);
    return cn_split_bf16_padded(w, rows, cols, transpose, n_padded, planes, reinterpret_cast<uint16_t *>(planes) + n, (void *)st);
}
int rn_gemm3(int act, int M, int N, int K, const float *A, int lda, const float *planes, const float *bias, float *C, int ldc, hipStream_t st,
             const float *aux = nullptr, int ldaux = 0, int relu_from = 1 << 30)
{
    return cn_linear_fwd_act(M, N, K, A, lda, planes, reinterpret_cast<const uint16_t *>(planes) + (size_t)N * K, bias, act, aux, ldaux, relu_from, C, ldc, (void *)st);
}
struct RnFwdWs { size_t te, teb, wih, ac0, a2, c2, total; }; // forward workspace (floats): split planes of the five weights + the padded te bias
RnFwdWs rn_fwd_ws()
{
    RnFwdWs w{};
    size_t off = 0;
    auto carve = [&](size_t n) { size_t o = off; off += (n + 63) & ~size_t(63); return o; };
    w.te = carve(384 * 256); w.teb = carve(384); w.wih = carve(384 * 128); w.ac0 = carve(512 * 128); w.a2 = carve(256 * 256); w.c2 = carve(256 * 256);
    w.total = off;
    return w;
}
} // namespace

extern "C" int64_t cn_rn_seq_workspace_floats(int T, int N) { return (T > 0 && N > 0) ? (int64_t)rn_ws(T, N).total : 0; }
extern "C" int64_t cn_rn_seq_fwd_workspace_floats(void) { return (int64_t)rn_fwd_ws().total; }

static int rn_check(int T, int N, int H, const void *a, const void *b, const void *c, const void *d, const cn_rn_weights *w, const cn_rn_saved *sv)
{
    CN_REQUIRE(T >= 1 && N >= 1 && H >= 1 && H <= CN_MAX_HUMANS, "cn_rn_seq: bad shape T=%d N=%d H=%d", T, N, H);
    CN_REQUIRE(a && b && c && d && w && sv, "cn_rn_seq: null argument");
    const void *const *wp = reinterpret_cast<const void *const *>(w);
    for (size_t i = 0; i < sizeof(cn_rn_weights) / sizeof(void *); ++i) CN_REQUIRE(wp[i], "cn_rn_seq: weight pointer #%zu is null", i);
    const void *const *sp = reinterpret_cast<const void *const *>(sv);
    for (size_t i = 0; i < sizeof(cn_rn_saved) / sizeof(void *); ++i) CN_REQUIRE(sp[i], "cn_rn_seq: saved-activation pointer #%zu is null", i);
    return CN_OK;
}

// The weight preparation of one optimiser step's sequence as jobs of ONE grouped launch (cn_split_group_launch): forward planes (te: 320 rows
// padded to 384 = three 128-column tiles; its bias padded with zeros) and the transposed planes of the backward's dX products.
int rn_seq_prep_jobs(const cn_rn_weights *w, float *fwd_ws, float *bwd_ws, int T, int N, CnSplitJob *out)
{
    int n = 0;
    if (fwd_ws) {
        const RnFwdWs F = rn_fwd_ws();
        out[n++] = cn_split_job(w->te_w, 320, 256, 0, 384, fwd_ws + F.te);
        out[n++] = cn_split_job(w->wih, 384, 128, 0, 0, fwd_ws + F.wih);
        out[n++] = cn_split_job(w->ac0_w, 512, 128, 0, 0, fwd_ws + F.ac0);
        out[n++] = cn_split_job(w->a2_w, 256, 256, 0, 0, fwd_ws + F.a2);
        out[n++] = cn_split_job(w->c2_w, 256, 256, 0, 0, fwd_ws + F.c2);
        out[n++] = CnSplitJob{w->te_b, fwd_ws + F.teb, nullptr, 384, 0, 0, 320, 0};
    }
    if (bwd_ws) {
        const RnWs L = rn_ws(T, N);
        out[n++] = cn_split_job(w->a2_w, 256, 256, 1, 0, bwd_ws + L.a2T);
        out[n++] = cn_split_job(w->c2_w, 256, 256, 1, 0, bwd_ws + L.c2T);
        out[n++] = cn_split_job(w->ac0_w, 512, 128, 1, 0, bwd_ws + L.ac0T);
        out[n++] = cn_split_job(w->wih, 384, 128, 1, 0, bwd_ws + L.wihT);
        out[n++] = cn_split_job(w->edge_w, 64, 256, 1, 0, bwd_ws + L.edgeT);
        out[n++] = cn_split_job(w->te_w, 320, 256, 1, 0, bwd_ws + L.teT);
    }
    return n;
}

extern "C" int cn_rn_seq_fwd(int T, int N, int H, const float *robot_node, const float *temporal, const float *out_sp, const int *row_off, const float *h0,
                             const float *masks, const float *actions, const cn_rn_weights *w, const cn_rn_saved *sv, float *ws, float *value, float *logp,
                             void *stream)
{
    return rn_seq_fwd_impl(T, N, H, robot_node, temporal, out_sp, row_off, h0, masks, actions, w, sv, ws, value, logp, stream, false);
}

int rn_seq_fwd_impl(int T, int N, int H, const float *robot_node, const float *temporal, const float *out_sp, const int *row_off, const float *h0,
                    const float *masks, const float *actions, const cn_rn_weights *w, const cn_rn_saved *sv, float *ws, float *value, float *logp,
                    void *stream, bool prepared)
{
    if (int rc = cn_require_device()) return rc;
    if (int rc = rn_check(T, N, H, robot_node, temporal, out_sp, row_off, w, sv)) return rc;
    CN_REQUIRE(h0 && masks && actions && ws && value && logp, "cn_rn_seq_fwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int B = T * N;
    const RnFwdWs F = rn_fwd_ws();
    int rc;
    if (!prepared) { // split planes of this optimiser step's weights, one grouped launch
        CnSplitJob jobs[8];
        const int nj = rn_seq_prep_jobs(w, ws, nullptr, T, N, jobs);
        if ((rc = cn_split_group_launch(jobs, nj, st))) return rc;
    }
    hipLaunchKernelGGL(robot_embed_kernel, dim3(B < 2048 ? B : 2048), dim3(256), 0, st, B, temporal, robot_node, w->rl_w, w->rl_b, sv->rs);
    CN_CHECK_LAUNCH();
    // z = [u (256) | relu(enc) (64) | .] in one product (both read robot_states), then the attention over the compacted rows, then edge -> z[320:384]
    // (the padded product writes zeros into z[:, 320:384]; the edge layer below overwrites them)
    if ((rc = rn_gemm3(ACT_NONE, B, 384, 256, sv->rs, 256, ws + F.te, ws + F.teb, sv->z, 384, st, nullptr, 0, 256))) return rc;
    hipLaunchKernelGGL(hr_attention_kernel, dim3((B + 3) / 4), dim3(256), 0, st, B, H, sv->z, 384, out_sp, row_off, sv->hr, sv->attn);
    CN_CHECK_LAUNCH();
    if ((rc = rn_gemm<ACT_RELU>(B, 64, 256, sv->hr, 256, w->edge_w, w->edge_b, sv->z + 320, 384, st))) return rc;
    if ((rc = rn_gemm3(ACT_NONE, B, 384, 128, sv->z + 256, 384, ws + F.wih, w->bih, sv->gi, 384, st))) return rc;
    if ((rc = cn_gru_seq_fwd(T, N, sv->gi, h0, masks, w->whh, w->bhh, sv->hs, sv->hms, sv->gates, stream))) return rc;
    if ((rc = rn_gemm3(ACT_TANH, B, 512, 128, sv->hs, 128, ws + F.ac0, w->ac0_b, sv->a1, 512, st))) return rc;
    if ((rc = rn_gemm3(ACT_TANH, B, 256, 256, sv->a1, 512, ws + F.a2, w->a2_b, sv->a2, 512, st))) return rc;
    if ((rc = rn_gemm3(ACT_TANH, B, 256, 256, sv->a1 + 256, 512, ws + F.c2, w->c2_b, sv->a2 + 256, 512, st))) return rc;
    hipLaunchKernelGGL(rn_head_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, B, sv->a2, w->cl_w, w->cl_b, w->fm_w, w->fm_b, w->logstd, actions, value, logp);
    CN_CHECK_LAUNCH();
    return CN_OK;
}

// side != NULL: the eight weight-gradient products (leaves of the dependency graph) go to that stream, each behind an event recorded on the
// main stream after the kernel that produces its dY; the dX chain, the GRU and the attention backward stay on `stream`.  The caller joins
// (waits for the side stream) before anything reads the gradients or reuses the workspace.  ev: five events.
extern "C" int cn_rn_seq_bwd(int T, int N, int H, const float *robot_node, const float *temporal, const float *out_sp, const int *row_off, const float *masks,
                             const float *actions, const cn_rn_weights *w, const cn_rn_saved *sv, const float *d_value, const float *d_logp, float *ws,
                             float *d_out_sp, float *d_h0, const cn_rn_grads *g, void *stream)
{
    return rn_seq_bwd_impl(T, N, H, robot_node, temporal, out_sp, row_off, masks, actions, w, sv, d_value, d_logp, ws, d_out_sp, d_h0, g, stream, nullptr, nullptr,
                           false, nullptr);
}

int rn_seq_bwd_impl(int T, int N, int H, const float *robot_node, const float *temporal, const float *out_sp, const int *row_off, const float *masks,
                    const float *actions, const cn_rn_weights *w, const cn_rn_saved *sv, const float *d_value, const float *d_logp, float *ws,
                    float *d_out_sp, float *d_h0, const cn_rn_grads *g, void *stream, hipStream_t side, hipEvent_t *ev, bool prepared, float **packed_heads)
{
    if (int rc = cn_require_device()) return rc;
    if (int rc = rn_check(T, N, H, robot_node, temporal, out_sp, row_off, w, sv)) return rc;
    CN_REQUIRE(masks && actions && d_value && d_logp && ws && d_out_sp && d_h0 && g, "cn_rn_seq_bwd: null argument");
    {
        const void *const *gp = reinterpret_cast<const void *const *>(g);
        for (size_t i = 0; i < sizeof(cn_rn_grads) / sizeof(void *); ++i) CN_REQUIRE(gp[i], "cn_rn_seq_bwd: gradient pointer #%zu is null", i);
    }
    hipStream_t st = (hipStream_t)stream;
    const int B = T * N;
    const RnWs L = rn_ws(T, N);
    float *d2 = ws + L.d2, *d1 = ws + L.d1, *dhs = ws + L.dhs, *dgi = ws + L.dgi, *dgh = ws + L.dgh, *dz = ws + L.dz, *dhr = ws + L.dhr, *drs = ws + L.drs;
    int rc;
    hipStream_t wst = side ? side : st; // the stream of the weight-gradient products
    int evi = 0;
    auto fork = [&]() -> int { // what `st` has enqueued so far precedes what `wst` gets from here on
        if (!side) return CN_OK;
        CN_HIP(hipEventRecord(ev[evi], st));
        CN_HIP(hipStreamWaitEvent(side, ev[evi], 0));
        ++evi;
        return CN_OK;
    };
    // ---- heads + the tanh of the second trunk layers; the heads' own weight gradients ----
    {
        const int blocks = B < 4 * L.small_rows ? (B + 3) / 4 : L.small_rows;
        hipLaunchKernelGGL(rn_head_bwd_kernel, dim3(blocks), dim3(256), 0, st, B, sv->a2, w->cl_w, w->fm_w, w->fm_b, w->logstd, actions, d_value, d_logp, d2, ws + L.small);
        CN_CHECK_LAUNCH();
        // the reduced head gradients: RN_HEAD_COLS floats in the bias-partial region (free until the first weight gradient below) -- or, for a
        // caller that scatters them itself (packed_heads), at the end of the head partials' region, which nothing writes before robot_linear's
        // partials at the very end of this call
        float *red = packed_heads ? ws + L.small + (size_t)(L.small_rows - 1) * (RN_HEAD_COLS > 2560 ? RN_HEAD_COLS : 2560) + 1024 : ws + L.dbp;
        hipLaunchKernelGGL(rn_reduce_rows_kernel, dim3((RN_HEAD_COLS + 15) / 16), dim3(256), 0, st, blocks, RN_HEAD_COLS, ws + L.small, red, nullptr, 0);
        CN_CHECK_LAUNCH();
        if (packed_heads) *packed_heads = red;
        else {
            CN_HIP(hipMemcpyAsync(g->fm_w, red, 512 * sizeof(float), hipMemcpyDeviceToDevice, st));
            CN_HIP(hipMemcpyAsync(g->cl_w, red + 512, 256 * sizeof(float), hipMemcpyDeviceToDevice, st));
            CN_HIP(hipMemcpyAsync(g->fm_b, red + 768, 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
            CN_HIP(hipMemcpyAsync(g->cl_b, red + 770, 1 * sizeof(float), hipMemcpyDeviceToDevice, st));
            CN_HIP(hipMemcpyAsync(g->logstd, red + 771, 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    }
    // ---- split planes of the transposed weights for the dX products (dX = dY W as an NT product with W^T), one grouped launch ----
    if (!prepared) {
        CnSplitJob jobs[8];
        const int nj = rn_seq_prep_jobs(w, nullptr, ws, T, N, jobs);
        if ((rc = cn_split_group_launch(jobs, nj, st))) return rc;
    }
    // ---- second trunk layers: weight gradients, then d1 = (d2 W2) (1 - a1^2) ----
    if ((rc = fork())) return rc; // (behind the head reduction and its copies: they read ws + L.dbp, which the first weight gradient overwrites)
    if ((rc = rn_wgrad(B, 256, 256, d2, 512, sv->a1, 512, ws, L, g->a2_w, g->a2_b, wst))) return rc;
    if ((rc = rn_wgrad(B, 256, 256, d2 + 256, 512, sv->a1 + 256, 512, ws, L, g->c2_w, g->c2_b, wst))) return rc;
    if ((rc = rn_gemm3(ACT_MUL_DTANH, B, 256, 256, d2, 512, ws + L.a2T, nullptr, d1, 512, st, sv->a1, 512))) return rc;
    if ((rc = rn_gemm3(ACT_MUL_DTANH, B, 256, 256, d2 + 256, 512, ws + L.c2T, nullptr, d1 + 256, 512, st, sv->a1 + 256, 512))) return rc;
    // ---- first trunk layers (output_linear folded in) ----
    if ((rc = fork())) return rc;
    if ((rc = rn_wgrad(B, 512, 128, d1, 512, sv->hs, 128, ws, L, g->ac0_w, g->ac0_b, wst))) return rc;
    if ((rc = rn_gemm3(ACT_NONE, B, 128, 512, d1, 512, ws + L.ac0T, nullptr, dhs, 128, st))) return rc;
    // ---- GRU over the sequence ----
    if ((rc = cn_gru_seq_bwd(T, N, sv->gates, sv->hms, masks, w->whh, dhs, dgi, dgh, d_h0, stream))) return rc;
    if ((rc = fork())) return rc;
    if ((rc = rn_wgrad(B, 384, 128, dgh, 384, sv->hms, 128, ws, L, g->whh, g->bhh, wst))) return rc;
    if ((rc = rn_wgrad(B, 384, 128, dgi, 384, sv->z + 256, 384, ws, L, g->wih, g->bih, wst))) return rc;
    // ---- d[enc | edge] = (dgi W_ih) relu'(.) -> dz[:, 256:384]; d hr = d edge W_e; attention backward; d u -> dz[:, 0:256] ----
    if ((rc = rn_gemm3(ACT_MUL_DRELU, B, 128, 384, dgi, 384, ws + L.wihT, nullptr, dz + 256, 384, st, sv->z + 256, 384))) return rc;
    if ((rc = fork())) return rc;
    if ((rc = rn_wgrad(B, 64, 256, dz + 320, 384, sv->hr, 256, ws, L, g->edge_w, g->edge_b, wst))) return rc;
    if ((rc = rn_gemm3(ACT_NONE, B, 256, 64, dz + 320, 384, ws + L.edgeT, nullptr, dhr, 256, st))) return rc;
    hipLaunchKernelGGL(hr_attention_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, B, H, sv->z, out_sp, row_off, sv->attn, dhr, dz, d_out_sp, 384, 384);
    CN_CHECK_LAUNCH();
    // ---- [u | enc] layer and robot_linear ----
    if ((rc = fork())) return rc;
    if ((rc = rn_wgrad(B, 320, 256, dz, 384, sv->rs, 256, ws, L, g->te_w, g->te_b, wst))) return rc;
    if ((rc = rn_gemm3(ACT_MUL_DRELU, B, 256, 320, dz, 384, ws + L.teT, nullptr, drs, 256, st, sv->rs, 256))) return rc;
    {
        const int blocks = B < 512 ? B : 512;
        hipLaunchKernelGGL(rn_rl_wgrad_kernel, dim3(blocks), dim3(256), 0, st, B, drs, temporal, robot_node, ws + L.small);
        CN_CHECK_LAUNCH();
        hipLaunchKernelGGL(rn_reduce_rows_kernel, dim3(2560 / 16), dim3(256), 0, st, blocks, 2560, ws + L.small, g->rl_w, g->rl_b, 1);
        CN_CHECK_LAUNCH();
    }
    return CN_OK;
}
