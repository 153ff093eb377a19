// rn_fused.hip -- the robot-node half of the policy forward as ONE kernel on gfx950 (everything after the human-human block):
//   robot_linear -> [u = Ws^T temporal_edge_layer(.) | encoder_linear] -> robot-human attention over out_sp ->
//   edge_attention_embed -> GRU cell (with the done mask) -> actor / critic trunks (output_linear folded in) -> critic_linear +
//   DiagGaussian head.               rl/networks/selfAttn_srnn_temp_node.py:395-449, srnn_model.py:35-105, model.py:56-80
// As separate launches this is ~12 small kernels whose M = E products fill a fraction of the chip each and whose LDS-using
// blocks cannot even co-reside with the 160 KB workgroups of the fused human-human kernel; here one workgroup (8 wavefronts)
// owns 16 envs, keeps every per-env activation in LDS and walks the ~1.3 MB of weights once, straight from L2 into MFMA operand
// registers.  The products run in the same bf16x3 split precision as the human-human kernel (fp32 operand = hi + lo bf16, products
// hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulation): with exact fp32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the
// bf16 rate) the seven stages were bound by the fp32 matrix pipe -- in-kernel timers: 83 k of the kernel's 118 k cycles, at ~60 % pipe
// utilisation -- now they are bound by streaming the 1.3 MB weight image.
//
// Products run "transposed" (A operand = weight fragment, B operand = activations): the C layout then holds 4 consecutive output
// features of one env per lane, which is a float4 store into the next layer's [env][feature] LDS image; the B operand of the
// next product is two float4 reads of that image (k = 32 ks + 8*(lane>>4) + 0..7), split to bf16 hi / lo once per stage.
#include "rn_fused.h"
#include "row_plan.h"

#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TE = 16;     // envs per workgroup
// LDS activation images [16 envs][stride] floats; strides are K + 4 so that the 16 lanes of a float4 read hit distinct banks
constexpr int S512 = 516, S384 = 388, S128 = 132;
constexpr int O_R0 = 0;                     // 512: robot_states (256) -> hr (256) -> ac1 (512)
constexpr int O_R1 = O_R0 + TE * S512;      // 512: z = [u 256 | enc 64 | edge 64] -> ac2 (512)
constexpr int O_R2 = O_R1 + TE * S512;      // 384: gh
constexpr int O_R3 = O_R2 + TE * S384;      // 384: gi
constexpr int O_R4 = O_R3 + TE * S384;      // 128: h_in
constexpr int O_R5 = O_R4 + TE * S128;      // 128: h_new
constexpr int LDS_FLOATS = O_R5 + TE * S128;

__device__ __forceinline__ f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

enum { A_NONE = 0, A_RELU = 1, A_TANH = 2 };

// tanh(x) = 1 - 2 / (1 + e^{2x}) on v_exp_f32: absolute error < 3e-7 (e^{2x} overflows to inf -> 1, underflows to 0 -> -1)
__device__ __forceinline__ float fast_tanh(float x)
{
    const float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f); // 2 * log2(e)
    return 1.0f - 2.0f / (1.0f + e);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)); }

// The first k-steps' weight fragments of a stage, fetched BEFORE the barrier that releases the stage's input activations (they depend on
// nothing): the L2 round trip of every stage's first fragments (~1-2 us beside the ORCA tail, seven stages in a chain) then runs under the
// previous stage's epilogue / the barrier instead of behind it.
template <int K, int NFB, int D = 2> // D: k-steps fetched ahead (<= the stage's own prefetch depth - 1)
struct WPre {
    static constexpr int KS = K / 32, PF = KS < 3 ? KS : 3, N = (PF - 1 < D ? PF - 1 : D) > 0 ? (PF - 1 < D ? PF - 1 : D) : 1;
    bf16x8 ah[N][NFB], al[N][NFB];
};
template <int K, int NFB, int D>
__device__ __forceinline__ void wpre_load(WPre<K, NFB, D> &w, const float *__restrict__ Wfrag_, int fb_first, int fb_step, int lane)
{
    const bf16x8 *__restrict__ Wfrag = reinterpret_cast<const bf16x8 *>(Wfrag_);
    constexpr int KS = K / 32, PF = KS < 3 ? KS : 3;
#pragma unroll
    for (int p = 0; p < PF - 1 && p < D; ++p)
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
            const size_t base = ((size_t)(fb_first + j * fb_step) * KS + (p < KS ? p : KS - 1)) * 128 + lane;
            w.ah[p][j] = Wfrag[base]; w.al[p][j] = Wfrag[base + 64];
        }
}

// out[env][out_off + 16*fb + ..] = act(W[fb] . in[env][in_off ..] + bias) for the feature blocks fb = fb0 + wave, fb0 + wave + 4, ...
// Wfrag: baked fragments [fb][K/32][plane hi,lo][64 lanes][8 bf16]; NFB = feature blocks of this wavefront
template <int K, int NFB, int ACT, bool PRE = false, int D = 2, bool LEAN = false>
__device__ __forceinline__ void stage(const float *__restrict__ Wfrag_, int fb_first, int fb_step, const float *__restrict__ bias, const float *in, int in_stride,
                                      float *out, int out_stride, int out_off, int relu_from, int lane, const WPre<K, NFB, D> &pre = WPre<K, NFB, D>{})
{
    const bf16x8 *__restrict__ Wfrag = reinterpret_cast<const bf16x8 *>(Wfrag_);
    const int i = lane & 15, g = lane >> 4;
    constexpr int KS = K / 32;
    // activations of env i, k = 32 ks + 8 g .. + 7, as bf16 hi / lo (the B operand of every feature block of this stage); LEAN: split per
    // k-step inside the loop instead of up front (64 registers less at K = 256)
    bf16x8 bh[LEAN ? 1 : KS], bl[LEAN ? 1 : KS];
    auto split = [&](int ks, bf16x8 &h, bf16x8 &l) {
        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(in + i * in_stride + 32 * ks + 8 * g);
        const f32x4 x1 = *reinterpret_cast<const f32x4 *>(in + i * in_stride + 32 * ks + 8 * g + 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const __bf16 h0 = (__bf16)x0[u], h1 = (__bf16)x1[u];
            h[u] = h0; h[4 + u] = h1;
            l[u] = (__bf16)(x0[u] - (float)h0); l[4 + u] = (__bf16)(x1[u] - (float)h1);
        }
    };
#pragma unroll
    for (int ks = 0; ks < (LEAN ? 0 : KS); ++ks) {
        const f32x4 x0 = *reinterpret_cast<const f32x4 *>(in + i * in_stride + 32 * ks + 8 * g);
        const f32x4 x1 = *reinterpret_cast<const f32x4 *>(in + i * in_stride + 32 * ks + 8 * g + 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const __bf16 h0 = (__bf16)x0[u], h1 = (__bf16)x1[u];
            bh[ks][u] = h0; bh[ks][4 + u] = h1;
            bl[ks][u] = (__bf16)(x0[u] - (float)h0); bl[ks][4 + u] = (__bf16)(x1[u] - (float)h1);
        }
    }
    f32x4 acc[NFB];
#pragma unroll
    for (int j = 0; j < NFB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weight fragments: PF k-steps in flight (the scheduling barriers pin the issue points, see hh_fused.hip)
    constexpr int PF = KS < (LEAN ? 2 : 3) ? KS : (LEAN ? 2 : 3);
    bf16x8 ah[PF][NFB], al[PF][NFB];
#pragma unroll
    for (int p = 0; p < PF - 1; ++p)
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
            if (PRE && p < D) { ah[p][j] = pre.ah[p < D ? p : 0][j]; al[p][j] = pre.al[p < D ? p : 0][j]; continue; }
            const size_t base = ((size_t)(fb_first + j * fb_step) * KS + (p < KS ? p : KS - 1)) * 128 + lane;
            ah[p][j] = Wfrag[base]; al[p][j] = Wfrag[base + 64];
        }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { // fully unrolled: bh / bl must be statically indexed (a runtime index sends the arrays to scratch)
        const int kp = ks + PF - 1 < KS ? ks + PF - 1 : KS - 1;
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
            const size_t base = ((size_t)(fb_first + j * fb_step) * KS + kp) * 128 + lane;
            ah[(ks + PF - 1) % PF][j] = Wfrag[base]; al[(ks + PF - 1) % PF][j] = Wfrag[base + 64];
        }
        if (LEAN) split(ks, bh[0], bl[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
            acc[j] = mfma32(al[ks % PF][j], bh[LEAN ? 0 : ks], acc[j]);
            acc[j] = mfma32(ah[ks % PF][j], bl[LEAN ? 0 : ks], acc[j]);
            acc[j] = mfma32(ah[ks % PF][j], bh[LEAN ? 0 : ks], acc[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NFB; ++j) {
        const int f0 = (fb_first + j * fb_step) * 16 + 4 * g;
        f32x4 v = acc[j];
        if (bias) v += *reinterpret_cast<const f32x4 *>(bias + f0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ACT == A_RELU) { if (f0 + q >= relu_from) v[q] = fmaxf(v[q], 0.0f); }
            if (ACT == A_TANH) v[q] = fast_tanh(v[q]);
        }
        *reinterpret_cast<f32x4 *>(out + i * out_stride + out_off + f0) = v;
    }
}

// LEAN (the default since round 6; CN_RN_LEAN=0: the 255-register variant): the same chain -- same products, same order, same bits -- in 128
// registers per lane (activations split per k-step, two k-steps of weight fragments in flight, the wide stages in chunks of <= 3 feature blocks,
// nothing held across the barriers).  On its own it is a few us slower; but a workgroup then takes HALF of its CU's registers, and the ORCA tail's
// wavefronts (orca_lp3_kernel on the simulator's side stream: 32 registers, no LDS, 37 us of VALU work for the whole chip) run on the same CUs
// beside this latency chain instead of making it wait for CUs of its own: hh_fused -> rn_fused gap 17.7 -> 9.9 us, step 0.2806 -> 0.2705 ms at
// 4096 envs x 20 humans (same box), configs[4] 2.07 -> 2.05 ms beside the cooperative ORCA kernel.
template <bool LEAN>
__global__ __launch_bounds__(512, LEAN ? 4 : 2) void rn_fused_kernel(int E, int H, RnFusedArgs a)
{
    const CnStampScope stamp_scope(a.stamp);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __builtin_amdgcn_s_setprio(3); // critical path of the step: win the issue arbitration against the side stream's simulator wavefronts
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e0 = blockIdx.x * TE;
    float *R0 = smem + O_R0, *R1 = smem + O_R1, *R2 = smem + O_R2, *R3 = smem + O_R3, *R4 = smem + O_R4, *R5 = smem + O_R5;
    // ---- robot_linear.0: relu(W [256,9] . [temporal_edges(2) | robot_node(7)] + b); thread = feature; h_in -> LDS ----
    const int w4 = wave & 3, hi = wave >> 2; // two groups of four wavefronts run independent products side by side
    constexpr bool PRE = !LEAN;
    WPre<256, 5, 1> p_te; WPre<128, 6, 1> p_whh;
    WPre<256, 1> p_edge; WPre<128, 3> p_wih; WPre<128, 4> p_ac0; WPre<256, 4> p_2;
    if (PRE) { if (hi == 0) wpre_load(p_te, a.f_te, w4, 4, lane); else wpre_load(p_whh, a.f_whh, w4, 4, lane); }
    {
        const int n = tid & 255; // feature; the two thread halves split the envs
        float w[9];
#pragma unroll
        for (int d = 0; d < 9; ++d) w[d] = a.rl_w[n * 9 + d];
        const float bn = a.rl_b[n];
        for (int i = (tid >> 8) * (TE / 2); i < (tid >> 8) * (TE / 2) + TE / 2; ++i) {
            const int e = e0 + i < E ? e0 + i : E - 1; // the tail workgroup repeats the last env (results discarded)
            float acc = bn;
            acc += a.temporal[e * 2] * w[0];
            acc += a.temporal[e * 2 + 1] * w[1];
#pragma unroll
            for (int d = 0; d < 7; ++d) acc += a.robot_node[e * 7 + d] * w[2 + d];
            acc = fmaxf(acc, 0.0f);
            R0[i * S512 + n] = acc;
            if (a.tap_robot && e0 + i < E) a.tap_robot[(size_t)(e0 + i) * 256 + n] = acc;
        }
        for (int idx = tid; idx < TE * 128; idx += 512) {
            const int i = idx >> 7, c = idx & 127;
            const int e = e0 + i < E ? e0 + i : E - 1;
            R4[i * S128 + c] = a.hxs_in[(size_t)e * 128 + c];
        }
    }
    __syncthreads();
    // ---- z = [u (256) | relu(enc) (64)] = te_w [320,256] . robot_states + te_b ;  gh = W_hh [384,128] . h_in (unmasked, no bias) ----
    if (hi == 0 && LEAN) { // (five feature blocks do not fit 128 registers: three and two)
        stage<256, 3, A_RELU, false, 2, true>(a.f_te, w4, 4, a.te_b, R0, S512, R1, S512, 0, 256, lane);
        stage<256, 2, A_RELU, false, 2, true>(a.f_te, w4 + 12, 4, a.te_b, R0, S512, R1, S512, 0, 256, lane);
    } else if (hi == 0) stage<256, 5, A_RELU, PRE, 1, LEAN>(a.f_te, w4, 4, a.te_b, R0, S512, R1, S512, 0, 256, lane, p_te);
    else if (!LEAN) stage<128, 6, A_NONE, PRE, 1, LEAN>(a.f_whh, w4, 4, nullptr, R4, S128, R2, S384, 0, 0, lane, p_whh);
    else { // (six feature blocks of weight fragments, two k-steps deep, do not fit 128 registers: three and three)
        stage<128, 3, A_NONE, false, 2, true>(a.f_whh, w4, 4, nullptr, R4, S128, R2, S384, 0, 0, lane);
        stage<128, 3, A_NONE, false, 2, true>(a.f_whh, w4 + 12, 4, nullptr, R4, S128, R2, S384, 0, 0, lane);
    }
    // the next two products' first fragments travel while the attention runs
    if (PRE) {
        if (hi == 0) wpre_load(p_edge, a.f_edge, w4, 4, lane);
        wpre_load(p_wih, a.f_wih, wave, 8, lane);
    }
    __syncthreads();
    // ---- robot-human attention (u-form, see hr_attention_kernel in policy.hip): wavefront w owns envs 2w, 2w+1 ----
    // out_sp was written a moment ago by the human-human kernel on (mostly) other XCDs: every row read is a trip to the fabric.  A
    // row-by-row loop is a chain of such trips (two passes x nd rows x 2 envs: ~25 of them, the largest single item of this kernel),
    // so the first 8 rows of BOTH envs are fetched up front into registers; envs with more rows walk the rest in chunks of 8.
    {
        constexpr int CH = 8;
        int r0q[2], ndq[2];
        float x[2][CH][4];
        const int *row_off = a.row_off; // written by the human-human kernel in front of this one: the plan's offsets or its own scan
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = 2 * wave + q;
            const int e = e0 + i < E ? e0 + i : E - 1;
            r0q[q] = row_off[e]; ndq[q] = row_off[e + 1] - r0q[q];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float *row = a.out_sp + (size_t)(r0q[q] + (u < ndq[q] ? u : ndq[q] - 1)) * 256;
                x[q][u][0] = row[lane]; x[q][u][1] = row[64 + lane]; x[q][u][2] = row[128 + lane]; x[q][u][3] = row[192 + lane];
            }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = 2 * wave + q;
            const int r0 = r0q[q], nd = ndq[q];
            const float *ue = R1 + i * S512;
            const float u0 = ue[lane], u1 = ue[64 + lane], u2 = ue[128 + lane], u3 = ue[192 + lane];
            float s = -INFINITY;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float tot = wv_sum(u0 * x[q][u][0] + u1 * x[q][u][1] + u2 * x[q][u][2] + u3 * x[q][u][3]);
                if (lane == u && u < nd) s = tot * ((float)H / 8.0f);
            }
            for (int j0 = CH; j0 < nd; j0 += CH) {
                float y[CH][4];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float *row = a.out_sp + (size_t)(r0 + (j0 + u < nd ? j0 + u : nd - 1)) * 256;
                    y[u][0] = row[lane]; y[u][1] = row[64 + lane]; y[u][2] = row[128 + lane]; y[u][3] = row[192 + lane];
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float tot = wv_sum(u0 * y[u][0] + u1 * y[u][1] + u2 * y[u][2] + u3 * y[u][3]);
                    if (lane == j0 + u && j0 + u < nd) s = tot * ((float)H / 8.0f);
                }
            }
            const float mx = wv_max(s);
            const float p = lane < nd ? expf(s - mx) : 0.0f;
            const float denom = wv_sum(p);
            const float at = p / denom;
            if (a.tap_attn && lane < H && e0 + i < E) a.tap_attn[(size_t)(e0 + i) * H + lane] = at;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float aj = u < nd ? wv_readlane(at, u) : 0.0f;
                if (u < nd) { o0 += aj * x[q][u][0]; o1 += aj * x[q][u][1]; o2 += aj * x[q][u][2]; o3 += aj * x[q][u][3]; }
            }
            for (int j0 = CH; j0 < nd; j0 += CH) {
                float y[CH][4];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float *row = a.out_sp + (size_t)(r0 + (j0 + u < nd ? j0 + u : nd - 1)) * 256;
                    y[u][0] = row[lane]; y[u][1] = row[64 + lane]; y[u][2] = row[128 + lane]; y[u][3] = row[192 + lane];
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float aj = wv_readlane(at, j0 + u < nd ? j0 + u : 0);
                    if (j0 + u < nd) { o0 += aj * y[u][0]; o1 += aj * y[u][1]; o2 += aj * y[u][2]; o3 += aj * y[u][3]; }
                }
            }
            float *o = R0 + i * S512; // robot_states are dead: hr takes their place
            o[lane] = o0; o[64 + lane] = o1; o[128 + lane] = o2; o[192 + lane] = o3;
            if (a.tap_hr && e0 + i < E) {
                float *t = a.tap_hr + (size_t)(e0 + i) * 256;
                t[lane] = o0; t[64 + lane] = o1; t[128 + lane] = o2; t[192 + lane] = o3;
            }
        }
    }
    __syncthreads();
    // ---- edge = relu(edge_attention_embed [64,256] . hr + b) -> z[320:384] ----
    if (hi == 0) stage<256, 1, A_RELU, PRE, 2, LEAN>(a.f_edge, w4, 4, a.edge_b, R0, S512, R1, S512, 320, 0, lane, p_edge);
    if (PRE) wpre_load(p_ac0, a.f_ac0, wave, 8, lane);
    __syncthreads();
    // ---- gi = W_ih [384,128] . [enc | edge] + b_ih ----
    stage<128, 3, A_NONE, PRE, 2, LEAN>(a.f_wih, wave, 8, a.bih, R1 + 256, S512, R3, S384, 0, 0, lane, p_wih);
    if (PRE) { if (hi == 0) wpre_load(p_2, a.f_a2, w4, 4, lane); else wpre_load(p_2, a.f_c2, w4, 4, lane); }
    __syncthreads();
    // ---- GRU cell, pointwise part (gate order r,z,n; h and gh masked by the done mask: srnn_model.py:43-46) ----
    for (int idx = tid; idx < TE * 128; idx += 512) {
        const int i = idx >> 7, c = idx & 127;
        const int e = e0 + i < E ? e0 + i : E - 1;
        const float m = a.masks[e];
        const float *gie = R3 + i * S384, *ghe = R2 + i * S384;
        const float hr = m * ghe[c] + a.bhh[c], hz = m * ghe[128 + c] + a.bhh[128 + c], hn = m * ghe[256 + c] + a.bhh[256 + c];
        const float r = fast_sigmoid(gie[c] + hr);
        const float z = fast_sigmoid(gie[128 + c] + hz);
        const float n = fast_tanh(gie[256 + c] + r * hn);
        const float h = m * R4[i * S128 + c];
        const float hnew = (1.0f - z) * n + z * h;
        R5[i * S128 + c] = hnew;
        if (e0 + i < E) a.hxs_out[(size_t)(e0 + i) * 128 + c] = hnew;
    }
    __syncthreads();
    // ---- actor / critic trunks: tanh((W0 Wo) h + ..) [512,128], then the two [256,256] second layers ----
    stage<128, 4, A_TANH, PRE, 2, LEAN>(a.f_ac0, wave, 8, a.ac0_b, R5, S128, R0, S512, 0, 0, lane, p_ac0);
    __syncthreads();
    if (LEAN) {
        const float *fw = hi == 0 ? a.f_a2 : a.f_c2, *fb = hi == 0 ? a.a2_b : a.c2_b;
        const float *src = hi == 0 ? R0 : R0 + 256;
        stage<256, 2, A_TANH, false, 2, true>(fw, w4, 4, fb, src, S512, R1, S512, hi * 256, 0, lane);
        stage<256, 2, A_TANH, false, 2, true>(fw, w4 + 8, 4, fb, src, S512, R1, S512, hi * 256, 0, lane);
    } else if (hi == 0) stage<256, 4, A_TANH, PRE, 2, LEAN>(a.f_a2, w4, 4, a.a2_b, R0, S512, R1, S512, 0, 0, lane, p_2);
    else stage<256, 4, A_TANH, PRE, 2, LEAN>(a.f_c2, w4, 4, a.c2_b, R0 + 256, S512, R1, S512, 256, 0, lane, p_2);
    __syncthreads();
    // ---- critic_linear + DiagGaussian head (model.py:64-72): wavefront w owns envs 2w, 2w+1 ----
    for (int q = 0; q < 2; ++q) {
        const int i = 2 * wave + q;
        if (e0 + i >= E) break;
        const int e = e0 + i;
        const float *av = R1 + i * S512, *cv = av + 256;
        float sv = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = lane + 64 * k;
            sv += cv[d] * a.cl_w[d];
            s0 += av[d] * a.fm_w[d];
            s1 += av[d] * a.fm_w[256 + d];
            if (a.tap_actor) a.tap_actor[(size_t)e * 256 + d] = av[d];
        }
        sv = wv_sum(sv); s0 = wv_sum(s0); s1 = wv_sum(s1);
        if (lane == 0) {
            a.value[e] = sv + a.cl_b[0];
            if (a.action) {
                const float mean0 = s0 + a.fm_b[0], mean1 = s1 + a.fm_b[1];
                const float ls0 = a.logstd[0], ls1 = a.logstd[1];
                const float sd0 = expf(ls0), sd1 = expf(ls1);
                const float a0 = a.eps ? mean0 + sd0 * a.eps[2 * e] : mean0;
                const float a1 = a.eps ? mean1 + sd1 * a.eps[2 * e + 1] : mean1;
                a.action[2 * e] = a0; a.action[2 * e + 1] = a1;
                const float HALF_LOG_2PI = 0.91893853320467274178f;
                const float d0 = a0 - mean0, d1 = a1 - mean1;
                a.logp[e] = (-(d0 * d0) / (2.0f * sd0 * sd0) - ls0 - HALF_LOG_2PI) + (-(d1 * d1) / (2.0f * sd1 * sd1) - ls1 - HALF_LOG_2PI);
            }
        }
    }
}

// fp32 row-major W [N,K] -> [fb = N/16][K/32][plane hi,lo][64 lanes][8 bf16]: lane (f, kk) holds W[16 fb + f][32 ks + 8 kk .. +7]
__global__ void rn_bake_kernel(int N, int K, const float *__restrict__ w, __bf16 *__restrict__ out)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // one weight
    if (idx >= (size_t)N * K) return;
    const int u = idx & 7, lane = (idx >> 3) & 63;
    const size_t rest = idx >> 9;
    const int KS = K / 32;
    const int ks = (int)(rest % KS), fb = (int)(rest / KS);
    const float x = w[(size_t)(fb * 16 + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4) + u];
    const __bf16 hi = (__bf16)x;
    const size_t base = (((size_t)fb * KS + ks) * 2) * 512 + lane * 8 + u;
    out[base] = hi;
    out[base + 512] = (__bf16)(x - (float)hi);
}

} // namespace

int rn_fused_bake(int N, int K, const float *w, float *out, hipStream_t st)
{
    CN_REQUIRE(N % 16 == 0 && K % 32 == 0, "rn_fused_bake: N must be a multiple of 16, K of 32");
    const size_t n = (size_t)N * K;
    hipLaunchKernelGGL(rn_bake_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, N, K, w, reinterpret_cast<__bf16 *>(out));
    CN_CHECK_LAUNCH();
    return CN_OK;
}

int rn_fused_forward(int E, int H, const RnFusedArgs &args, hipStream_t st)
{
    constexpr size_t lds = (size_t)LDS_FLOATS * sizeof(float);
    static thread_local int attr_dev = -1;
    int dev = 0;
    CN_HIP(hipGetDevice(&dev));
    if (dev != attr_dev) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rn_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rn_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_dev = dev;
    }
    RnFusedArgs a = args;
    a.stamp = cn_stamp_slot(CN_K_RN_FUSED);
    static const int lean = getenv("CN_RN_LEAN") ? atoi(getenv("CN_RN_LEAN")) : 1;
    if (lean) hipLaunchKernelGGL(rn_fused_kernel<true>, dim3((E + TE - 1) / TE), dim3(512), lds, st, E, H, a);
    else hipLaunchKernelGGL(rn_fused_kernel<false>, dim3((E + TE - 1) / TE), dim3(512), lds, st, E, H, a);
    CN_CHECK_LAUNCH();
    return CN_OK;
}
