// train_internal.h -- pieces of the training path shared between linear.hip, policy.hip and train_step.hip (not part of the C ABI).
#pragma once
#include "common.h"

// One weight-preparation job of a grouped launch: the bf16 hi / lo planes of w (or of w^T), zero-padded to Nw rows, in the fragment order
// cn_linear_fwd loads (cn_split_bf16_padded) -- or, with Kw == 0, a bias vector copied into an Nw-float buffer behind `hi`, zero-padded.
struct CnSplitJob {
    const float *w;
    void *hi, *lo;
    int Nw, Kw, transpose, Nreal; // the weight as the product sees it: [Nreal rows (padded to Nw), Kw]
    int block0;                   // filled by the launcher
};
constexpr int CN_SPLIT_MAX_JOBS = 20;
int cn_split_group_launch(CnSplitJob *jobs, int n, hipStream_t st);                                            // linear.hip
// job for cn_split_bf16_padded(w, rows, cols, transpose, n_padded, planes) with hi = planes, lo = planes + n elements (policy.hip's rn_split)
CnSplitJob cn_split_job(const float *w, int rows, int cols, int transpose, int n_padded, float *planes);

// the split jobs of one optimiser step's robot-node sequence: five for cn_rn_seq_fwd (+ the padded te bias), six for cn_rn_seq_bwd
int rn_seq_prep_jobs(const cn_rn_weights *w, float *fwd_ws, float *bwd_ws, int T, int N, CnSplitJob *out);     // policy.hip: returns the count (12)
int rn_seq_fwd_impl(int T, int N, int H, const float *robot_node, const float *temporal, const float *out_sp, const int *row_off, const float *h0,
                    const float *masks, const float *actions, const cn_rn_weights *w, const cn_rn_saved *sv, float *ws, float *value, float *logp,
                    void *stream, bool prepared);
// side != NULL: the eight weight-gradient products go to that stream behind events recorded on `stream` (ev: five events); the caller joins.
// packed_heads != NULL: the heads' gradients stay packed there ([fc_mean.w 512 | critic_linear.w 256 | fc_mean.b 2 | critic_linear.b 1 | logstd 2])
// instead of five device-to-device copies; prepared: the transposed split planes are already in the workspace (rn_seq_prep_jobs)
int rn_seq_bwd_impl(int T, int N, int H, const float *robot_node, const float *temporal, const float *out_sp, const int *row_off, const float *masks,
                    const float *actions, const cn_rn_weights *w, const cn_rn_saved *sv, const float *d_value, const float *d_logp, float *ws,
                    float *d_out_sp, float *d_h0, const cn_rn_grads *g, void *stream, hipStream_t side, hipEvent_t *ev, bool prepared, float **packed_heads);
