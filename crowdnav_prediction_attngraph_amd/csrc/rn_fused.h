// rn_fused.h -- internal interface of the fused robot-node kernel (rn_fused.hip), used by policy.hip.
#pragma once
#include "common.h"

struct RnFusedArgs {
    // observation / recurrent inputs
    const float *temporal, *robot_node, *hxs_in, *masks, *eps;
    const float *out_sp;  // [live rows, 256] from the human-human block
    const int *row_off;   // [E + 1]
    // weights: baked MFMA fragments (rn_fused_bake) + fp32 biases / small vectors
    const float *rl_w, *rl_b;             // robot_linear.0 [256,9]
    const float *f_te, *te_b;             // [u = Ws^T Wt (256) ; encoder_linear (64)] [320,256]
    const float *f_whh, *bhh;             // GRU W_hh [384,128]
    const float *f_edge, *edge_b;         // edge_attention_embed [64,256]
    const float *f_wih, *bih;             // GRU W_ih [384,128]
    const float *f_ac0, *ac0_b;           // (actor.0 ; critic.0) ∘ output_linear [512,128]
    const float *f_a2, *a2_b, *f_c2, *c2_b; // actor.2, critic.2 [256,256]
    const float *cl_w, *cl_b, *fm_w, *fm_b, *logstd;
    // outputs
    float *value, *action, *logp, *hxs_out;
    // optional test taps (nullptr = not written)
    float *tap_robot, *tap_attn, *tap_hr, *tap_actor;
    unsigned long long *stamp; // launch stamps (common.h), filled in by rn_fused_forward
};

int rn_fused_bake(int N, int K, const float *w, float *out, hipStream_t st);
int rn_fused_forward(int E, int H, const RnFusedArgs &args, hipStream_t st);
