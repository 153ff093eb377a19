// hh_fused.h -- internal interface of the fused human-human block (hh_fused.hip), used by policy.hip.
#pragma once
#include "common.h"

// Weight images in MFMA-fragment streaming order (built once per weight snapshot by hh_fused_bake) + the fp32 pieces
struct HhFusedWeights {
    const void *emb2_frag; // [wave 4][ks 4][j 8][plane 2][lane 64][8 bf16]            256 KB
    const void *qkv_frag;  // [head 8][wave 4][ks 16][j 3 (q,k,v)][plane 2][64][8]      3 MB
    const void *os_frag;   // [head 8][wave 4][ks 2][j 4][plane 2][64][8]               512 KB
    const float *emb0_w, *emb0_b, *emb2_b, *qkv_b, *os_b;
    int prio; // raise the wavefront priority (s_setprio 3) against co-resident side-stream work
    // training forward (cn_hh_block_fwd): the activations the backward kernels need, written out while they pass through the registers
    // (all NULL in the rollout): e0 [rows,128] = relu(x W0^T + b0), x [rows,512] = relu(e0 W2^T + b2), qkv [rows,1536] (q NOT scaled),
    // attn [rows,512] = the attention output before out_proj.  qscale multiplies the scores (1 when the 1/sqrt(64) is folded into the
    // q weights, as cn_policy_set_weights does; 0.125 for the training weights, which are folded without it).
    float qscale;
    float *e0_out, *x_out, *qkv_out, *attn_out;
};

constexpr size_t HH_EMB2_FRAG_BYTES = (size_t)512 * 128 * 4;
constexpr size_t HH_QKV_FRAG_BYTES = (size_t)1536 * 512 * 4;
constexpr size_t HH_OS_FRAG_BYTES = (size_t)256 * 512 * 4;

// emb2_w [512,128], qkv_w [1536,512] (folded q|k|v), os_w [256,512] (folded out_proj∘spatial_linear): fp32 row-major, device
int hh_fused_bake(const float *emb2_w, const float *qkv_w, const float *os_w, void *emb2_frag, void *qkv_frag, void *os_frag, hipStream_t st);

// out_sp [row_off[E], 256] = relu(spatial_linear(out_proj(attention(...)))) on the compacted live rows.  det != NULL: the kernel first
// builds row_off [E+1] (exclusive prefix of clamp(detected_human_num, 1, H)) itself and leaves it behind for the caller's next kernels
// (live_total, optional, accumulates row_off[E]); det == NULL: row_off is an input.
// row_plan: cn_obs.row_plan of the observation (or NULL): when valid for this batch the kernel takes its row offsets and tile packing
// (row_plan.h) instead of scanning det and cutting consecutive envs (row_off still receives the offsets)
int hh_fused_forward(int E, int H, int D, const float *spatial_edges, const float *det, int *row_off, unsigned long long *live_total,
                     const HhFusedWeights &w, float *out_sp, hipStream_t st, const int32_t *row_plan = nullptr);
