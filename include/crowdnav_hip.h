/*
 * crowdnav_hip.h -- C ABI of libcrowdnav_hip.so, the MI355X (gfx950) implementation of the CrowdNav++ hot path.
 *
 * The reference (Shuijing725/CrowdNav_Prediction_AttnGraph) is pure Python; its only native boundary on this path is
 * the third-party `rvo2` module.  There is therefore no existing FFI to bind to: this header defines the boundary a
 * maintainer would bind from the reference's Python (ctypes stub in INTEGRATION.md), one entry point per reference
 * interface it replaces:
 *
 *   cn_env_create/destroy      <- rl/networks/envs.py:97-140 make_vec_envs (+ :36-94 make_env: thisSeed = seed + rank,
 *                                 nenv, phase) and crowd_sim/envs/crowd_sim.py:88-202 configure()
 *   cn_env_reset               <- VecEnv.reset(): rl/networks/shmem_vec_env.py:74-79 -> crowd_sim_var_num.py:303-363
 *   cn_env_step                <- VecEnv.step(): shmem_vec_env.py:136-142 (auto-reset) -> crowd_sim_var_num.py:366-460
 *                                 / crowd_sim_pred.py:100-214, incl. ORCA humans (crowd_nav/policy/orca.py:64-117,
 *                                 i.e. rvo2.PyRVOSimulator.doStep) and bench.Monitor episode stats (envs.py:70-73)
 *   cn_orca_solve              <- rvo2: PyRVOSimulator.addAgent / setAgentPosition,Velocity,PrefVelocity / doStep / getAgentVelocity(0) (orca.py:80-114)
 *   cn_policy_create/destroy/set_weights <- rl/networks/model.py:16-46 Policy.__init__ / load_state_dict
 *   cn_policy_act              <- rl/networks/model.py:56-74 Policy.act (-> selfAttn_srnn_temp_node.py:360-449)
 *   cn_policy_get_value        <- rl/networks/model.py:76-80 Policy.get_value
 *   cn_gae                     <- rl/networks/storage.py:123-132 RolloutStorage.compute_returns (use_gae branch)
 *   cn_adv_stats / cn_adv_normalize <- rl/ppo/ppo.py:37-39 advantage normalisation (split so that N GPUs can
 *                                 all-reduce the three partial sums in between)
 *   cn_ppo_loss_fwd / cn_ppo_loss_bwd <- rl/ppo/ppo.py:66-84 clipped surrogate + clipped value loss (and their gradients)
 *   cn_adam_clip_step          <- rl/ppo/ppo.py:86-93 nn.utils.clip_grad_norm_ + torch.optim.Adam.step (ppo.py:32) over one flat bucket
 *   cn_linear_* / cn_hh_attention_* / cn_hr_attention_* / cn_gru_* / cn_embed0_*
 *                              <- the operators of Policy.evaluate_actions (rl/networks/model.py:82-90) as autograd runs
 *                                 them forward and backward inside PPO.update (rl/ppo/ppo.py:60-95)
 *   cn_env_get_danger_min_dist / cn_env_set_case_counters
 *                              <- test phase: Danger(min_dist) (crowd_sim_var_num.py:499-533) and the per-case seeding that
 *                                 rl/evaluation.py's protocol relies on (crowd_sim_var_num.py:316-318,337)
 *   cn_env_snapshot_bytes / cn_env_save / cn_env_load
 *                              <- (new) simulator state for a bit-exact --resume (train.py:105-108 restores the policy only)
 *   cn_gst_*                   <- gst_updated wrapper + VecPretextNormalize (CrowdSimPredRealGST-v0)
 *
 * Conventions: every function returns 0 on success and a negative cn_status otherwise (no C++ exception crosses the
 * ABI); cn_last_error() returns a thread-local message (the buffer is `thread_local`: concurrent callers on different
 * threads never see each other's text).  All tensor arguments are raw DEVICE pointers owned by the
 * caller (PyTorch allocates them); the library owns only the opaque handles (persistent per-env simulator state incl.
 * the numpy-compatible MT19937 streams, and the policy workspace / folded weights).  `stream` is a hipStream_t passed
 * as void*; all work is enqueued on it and nothing synchronises with the host.  A handle is re-entrant across handles,
 * not thread-safe on one handle.  There is no CPU fallback: every entry point fails if no gfx950 device is present.
 */
#ifndef CROWDNAV_HIP_H
#define CROWDNAV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CN_OK = 0,
    CN_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    CN_ERR_HIP = -2,       /* a HIP runtime call failed (see cn_last_error) */
    CN_ERR_NO_DEVICE = -3, /* no gfx950 device visible */
    CN_ERR_STATE = -4      /* call sequence error (e.g. step before reset, weights not set) */
} cn_status;

enum { CN_ENV_VARNUM = 0, CN_ENV_PRED = 1, CN_ENV_PRED_GST = 2, CN_ENV_COLLECT = 3 };   /* gym ids CrowdSimVarNum-v0 / CrowdSimPred-v0 /
                                                                                          * CrowdSimPredRealGST-v0 / CrowdSimVarNumCollect-v0 */
enum { CN_PHASE_TRAIN = 0, CN_PHASE_VAL = 1, CN_PHASE_TEST = 2 };
enum { CN_INFO_NOTHING = 0, CN_INFO_TIMEOUT = 1, CN_INFO_COLLISION = 2, CN_INFO_REACHGOAL = 3, CN_INFO_DANGER = 4 }; /* crowd_sim/envs/utils/info.py */

enum { CN_ROBOT_NETWORK = 0, CN_ROBOT_ORCA = 1, CN_ROBOT_SOCIAL_FORCE = 2 };
enum { CN_KIN_HOLONOMIC = 0, CN_KIN_UNICYCLE = 1 };   /* action_space.kinematics */
enum { CN_HUMANS_ORCA = 0, CN_HUMANS_SOCIAL_FORCE = 1 }; /* humans.policy */
#define CN_MAX_HUMANS 64 /* one wavefront lane per human */
#define CN_MAX_PRED 8

/* Mirrors the fields of crowd_nav/configs/config.py that the path reads (same names, same meaning). */
typedef struct {
    int32_t human_num;            /* sim.human_num */
    int32_t predict_steps;        /* sim.predict_steps */
    int32_t env_kind;             /* CN_ENV_* */
    int32_t randomize_attributes; /* env.randomize_attributes */
    int32_t random_goal_changing; /* humans.random_goal_changing */
    int32_t end_goal_changing;    /* humans.end_goal_changing */
    int32_t sort_humans;          /* args.sort_humans */
    int32_t phase;                /* CN_PHASE_TRAIN, CN_PHASE_TEST (seeds 1000 + case, 'truth' roll-out, future-zone Danger) or, CrowdSimPred-v0 only,
                                   * CN_PHASE_VAL (seeds 0 + case, Danger from the previous observation's predictions) */
    int32_t nenv;                 /* TOTAL number of envs across all GPUs: the case_counter stride (crowd_sim_var_num.py:348) */
    uint32_t val_size, test_size;
    int32_t robot_policy;         /* CN_ROBOT_NETWORK: cn_env_step's action drives the robot; CN_ROBOT_ORCA: robot.policy = 'orca'
                                   * (crowd_sim_var_num.py:371-375), ORCA on the robot's beliefs, the action argument is ignored;
                                   * CN_ROBOT_SOCIAL_FORCE: robot.policy = 'social_force' (crowd_nav/policy/social_force.py), likewise */
    int32_t robot_visible;        /* robot.visible: every human's ORCA / social force sees the robot as one more neighbour (crowd_sim.py:695-699);
                                   * human_num + human_num_range <= 63; not with CrowdSimPred-v0 + const_vel predictions (the reference fails there) */
    int32_t auto_reset;           /* 1 (default): vec-env semantics, a finished env is reset inside cn_env_step and `obs` holds the
                                   * reset observation (shmem_vec_env.py:139-142); 0: single gym env semantics
                                   * (crowd_sim_var_num.py:366-460 alone): the terminal observation is returned, cn_env_reset restarts */
    int32_t predict_truth;        /* CrowdSimPred-v0 with sim.predict_method = 'truth' (crowd_sim_pred.py:81, crowd_sim_var_num.py:180-206): the
                                   * observation carries the humans' TRUE future positions (their own ORCA rolled forward predict_steps
                                   * times from the state just reached) instead of constant-velocity extrapolations */
    int32_t max_placement_attempts; /* bound of the reference's UNBOUNDED rejection sampling of human positions / goals
                                   * (crowd_sim_var_num.py:116-146, crowd_sim.py:415-450): after this many attempts the last
                                   * candidate is accepted; 0 = 65536.  Dense randomised crowds have seeds where the reference
                                   * loop runs for minutes, and a batch waits for its slowest env.
                                   * SEMANTIC DEVIATION from the reference (which never gives up): an episode / goal change whose
                                   * loop reaches the bound places that human at a position that still violates the minimum
                                   * distance.  The C oracle has the same bound, so the bit-exact tests hold; none of the
                                   * reference-generated traces or shipped evaluation logs reaches it (20 humans: < 300 attempts
                                   * at worst); it matters for crowds of ~50 randomised humans (BASELINE configs[4]). */
    int32_t human_num_range;      /* sim.human_num_range: the crowd holds human_num - range .. human_num + range humans (drawn at reset,
                                   * changed every 5 s: crowd_sim_var_num.py:103-104, :404-437, crowd_sim_pred.py:165-190); observations
                                   * always have human_num + human_num_range rows, which must be <= CN_MAX_HUMANS */
    int32_t kinematics;           /* CN_KIN_* (action_space.kinematics); unicycle: network-driven robot; in CrowdSimPred-v0 / PredRealGST-v0 the command
                                   * passes through smooth_action's noisy wheel model (crowd_sim.py:315-358) */
    int32_t humans_policy;        /* CN_HUMANS_* (humans.policy) */
    int32_t pred_interval;        /* int(data.pred_timestep // env.time_step) (crowd_sim.py:180-181): prediction k lies k * pred_interval simulation steps
                                   * ahead -- const_vel offsets (crowd_sim_var_num.py:212); 'truth': predict_steps * pred_interval rolls of which every
                                   * pred_interval-th is kept (:181, :206).  0 = 1 (every shipped config); at most 16 */
    double time_step, time_limit;
    double success_reward, collision_penalty, discomfort_dist, discomfort_penalty_factor;
    double circle_radius, arena_size;
    double human_radius, human_v_pref;
    double robot_radius, robot_v_pref, sensor_range;
    double goal_change_chance, end_goal_change_chance;
    double orca_neighbor_dist, orca_safety_space, orca_time_horizon, orca_time_horizon_obst;
    double sf_A, sf_B, sf_KI;     /* config.sf.* (crowd_nav/policy/social_force.py) */
    double robot_fov, human_fov;  /* robot.FOV, humans.FOV in units of pi (crowd_sim.py:122-123, detect_visible :513-552); 2 = all round */
} cn_env_config;

/* CN_ENV_COLLECT (crowd_sim/envs/crowd_sim_var_num_collect.py, driven by collect_data.py): the dataset generator of the GST predictor.
 * spatial_edges is then pred_info [E,H,4] = (frame id, prediction id, absolute px, py of the robot's belief; +inf for humans the
 * robot does not see); reward is 0, an episode never ends, a robot at its goal draws a new one. */
/* Observation tensors exactly as VecPyTorch returns them (float32, contiguous):
 * robot_node [E,1,7], temporal_edges [E,1,2], spatial_edges [E,H,D] (D = 2 or 2*(predict_steps+1)),
 * detected_human_num [E,1], visible_masks [E,H] (uint8 0/1; may be NULL). */
typedef struct {
    float *robot_node;
    float *temporal_edges;
    float *spatial_edges;
    float *detected_human_num;
    uint8_t *visible_masks;
    int32_t *row_plan; /* optional (may be NULL), cn_row_plan_words(E) int32: derived data of detected_human_num that cn_env_reset / cn_env_step
                        * write beside the observation and cn_policy_act reads with it -- the row offsets of the compacted (env, human) rows
                        * and a packing of the envs into equally filled tiles for the fused human-human kernel (csrc/row_plan.h).  Valid only
                        * together with the observation it was written with.  Not part of the reference's observation: without it (NULL, or
                        * a config whose step does not build one) the policy derives the same row offsets itself and walks the envs in order.
                        * The buffer needs NO initialisation (any content, e.g. a fresh hipMalloc): every reset / step either writes a complete
                        * plan or clears word 0, and the builders' cross-workgroup counter lives in the batch's own memory. */
} cn_obs;

typedef struct cn_env_batch cn_env_batch;
typedef struct cn_policy cn_policy;

/* Bumped whenever a struct layout, a signature or the snapshot format changes (round 4: cn_obs.row_plan, cn_env_config.robot_fov /
 * human_fov, the profiling entry points, snapshot layout CNENV004); the ctypes binding refuses a library that reports another number. */
#define CN_ABI_VERSION 404
const char *cn_last_error(void);
int cn_version(void);
int cn_device_count(void);

void cn_env_config_default(cn_env_config *cfg);
/* num_envs envs on the current device; env i gets thisSeed = seed + first_env_index + i (global env index, so that
 * trajectories do not depend on how many GPUs the batch is sharded over). */
int cn_env_create(const cn_env_config *cfg, int num_envs, int64_t seed, int64_t first_env_index, cn_env_batch **out);
int cn_env_destroy(cn_env_batch *env);
int cn_env_obs_width(const cn_env_config *cfg);
/* Episodes are generated ahead of the reset that needs them (crowd_sim_var_num.py:303-363: seed, robot, humans by rejection sampling), on a
 * side stream beside the ORCA kernel.  One launch of that generator works for at most `ticks_10ns` x 10 ns per env and resumes in the next
 * step (default 4000 = 40 us): its wavefronts hold registers the policy's kernel, next on the caller's stream, needs.  The episodes do not
 * depend on the budget (0 = one human per launch); an env that resets before its next episode is complete generates it in place. */
int cn_env_set_pregen_budget(cn_env_batch *env, int64_t ticks_10ns);
/* Deferred tail.  A step leaves two pieces of work for its successor on the library's side stream: the ORCA programs the lane kernel could not
 * finish (a third of them: the linearProgram3 fallback) and the pre-generation of the next episodes.  By default cn_env_step enqueues them itself,
 * i.e. BEFORE whatever the caller enqueues next -- and when that is the policy's human-human kernel, which needs whole CUs, every wavefront of
 * the tail that got a CU first holds one of its workgroups back.  With deferral on, cn_env_step holds the tail back (configurations with the lane
 * kernel only) until cn_env_launch_tail(env, stream) -- to be called right after the big kernel went out on `stream`: the tail is ordered behind
 * that point and then runs beside the robot-node kernel.  cn_policy_set_post_hh_hook makes cn_policy_act call it at exactly that place.  A tail
 * nobody launched goes out with the next call that needs its results (cn_env_step, cn_env_reset, cn_env_save, the getters): results never depend
 * on the mode, only the timeline does. */
int cn_env_set_tail_deferral(cn_env_batch *env, int enabled);
int cn_env_launch_tail(cn_env_batch *env, void *stream);
/* int32 words of a cn_obs.row_plan buffer for a batch of num_envs envs */
int64_t cn_row_plan_words(int num_envs);
int cn_env_reset(cn_env_batch *env, const cn_obs *obs, void *stream);
/* actions [E,2] float32 (raw policy output; clipped inside like srnn.clip_action).  Outputs: reward [E] float32,
 * done [E] uint8, info [E] uint8 (CN_INFO_*), ep_return [E] float64 and ep_len [E] int32 (valid where done: the
 * bench.Monitor episode sum / length), not_done [E] float32 = 1 - done (the `masks` of train.py:185-186; may be NULL).
 * Envs that finish are reset in the same launch and `obs` holds the reset observation for them (shmem_vec_env.py:139-142). */
int cn_env_step(cn_env_batch *env, const float *actions, const cn_obs *obs, float *reward, uint8_t *done, uint8_t *info,
                double *ep_return, int32_t *ep_len, float *not_done, void *stream);
/* Orders everything the library has in flight on its internal side stream (ORCA of the current state, next-episode pre-generation)
 * before whatever is enqueued on `stream` next.  cn_env_step does this itself; a caller needs it before reading simulator state from
 * another stream, or to close a hipGraph capture of a block of steps (a capture may not end with unjoined work on a forked stream).
 * Experimental: the only in-tree graph caller is the probe tools/graph_probe.py (replay measured slower than eager launches). */
int cn_env_join(cn_env_batch *env, void *stream);
/* Debug/test access to the simulator state: copies humans [E,H,8] (px,py,vx,vy,gx,gy,radius,v_pref) and robot [E,8]
 * (px,py,vx,vy,gx,gy,theta,potential) as float64 into caller DEVICE buffers (either may be NULL). */
int cn_env_get_state(cn_env_batch *env, double *humans, double *robot, void *stream);
/* ORCA velocities of the humans for the CURRENT state (the ones the next cn_env_step will apply; they are computed
 * ahead of time on an internal side stream, overlapped with the caller's policy forward), [E,H,2] float32 */
int cn_env_get_human_actions(cn_env_batch *env, float *out, void *stream);
/* Danger(min_dist) of the last step (crowd_sim_var_num.py:499-533): in the test phase the smallest distance between the
 * robot and an intruded TRUE future position of a visible human (humans rolled forward predict_steps times with their
 * own ORCA policies, calc_human_future_traj('truth') :152-206); 0 for every other info and in the train phase.
 * out [E] float64 (device). */
int cn_env_get_danger_min_dist(cn_env_batch *env, double *out, void *stream);
/* len(self.humans) of every env: out [E] int32 (device).  Constant human_num unless sim.human_num_range > 0; the slots beyond it in
 * cn_env_get_state / cn_env_get_human_actions hold no human. */
int cn_env_get_human_counts(cn_env_batch *env, int32_t *out, void *stream);
/* Overwrite the per-env case counters (crowd_sim_var_num.py:316-318 `case_counter[phase] = test_case`, :337
 * rand_seed = offset[phase] + case_counter + thisSeed): the NEXT reset of env e generates the scenario of that case.
 * counters [E] uint64 (device).  Lets a batch replay chosen test cases (one per env) instead of consecutive ones. */
int cn_env_set_case_counters(cn_env_batch *env, const uint64_t *counters, void *stream);

/* Checkpointing of the simulator (the reference saves only the policy, train.py:213-219; resuming a run bit-exactly also needs
 * the env state: agent records, beliefs, case counters, private ORCA simulators, the numpy MT19937 streams).  A snapshot is an
 * opaque DEVICE buffer of cn_env_snapshot_bytes() bytes; cn_env_load only accepts a snapshot taken from a batch with the same
 * configuration, shape, seed and shard (first_env_index).  Both synchronise `stream` (they are not on the hot path). */
int64_t cn_env_snapshot_bytes(const cn_env_batch *env);
int cn_env_save(cn_env_batch *env, void *dst, void *stream);
int cn_env_load(cn_env_batch *env, const void *src, void *stream);

/* Stand-alone batched ORCA solve (the rvo2 replacement): B independent agents, each with n_other neighbours.
 * self [B,8] = px,py,vx,vy,radius,max_speed,pref_vx,pref_vy ; others [B,n_other,5] = px,py,vx,vy,radius (float32);
 * out_vel [B,2].  n_other <= 63.  Parameters as PyRVOSimulator.addAgent. */
int cn_orca_solve(int B, int n_other, const float *self, const float *others, float neighbor_dist, int max_neighbors,
                  float time_horizon, float time_step, float *out_vel, void *stream);

/* ---- policy (selfAttn_merge_srnn + DiagGaussian head) ---- */
/* Device pointers to the fp32 parameters, named after the reference state_dict keys (model.py / SURVEY.md 8a-P0). */
typedef struct {
    const float *robot_linear_w, *robot_linear_b;                 /* base.robot_linear.0 [256,9] */
    const float *emb0_w, *emb0_b;                                 /* base.spatial_attn.embedding_layer.0 [128,D] */
    const float *emb2_w, *emb2_b;                                 /* base.spatial_attn.embedding_layer.2 [512,128] */
    const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b;               /* base.spatial_attn.{q,k,v}_linear [512,512] */
    const float *in_proj_w, *in_proj_b;                           /* base.spatial_attn.multihead_attn.in_proj_* [1536,512] */
    const float *out_proj_w, *out_proj_b;                         /* ...multihead_attn.out_proj [512,512] */
    const float *spatial_linear_w, *spatial_linear_b;             /* base.spatial_linear.0 [256,512] */
    const float *attn_temporal_w, *attn_temporal_b;               /* base.attn.temporal_edge_layer.0 [64,256] */
    const float *attn_spatial_w, *attn_spatial_b;                 /* base.attn.spatial_edge_layer.0 [64,256] */
    const float *enc_w, *enc_b;                                   /* base.humanNodeRNN.encoder_linear [64,256] */
    const float *edge_embed_w, *edge_embed_b;                     /* base.humanNodeRNN.edge_attention_embed [64,256] */
    const float *gru_w_ih, *gru_w_hh, *gru_b_ih, *gru_b_hh;       /* base.humanNodeRNN.gru.* [384,128],[384,128],[384],[384] */
    const float *out_w, *out_b;                                   /* base.humanNodeRNN.output_linear [256,128] */
    const float *actor0_w, *actor0_b, *actor2_w, *actor2_b;       /* base.actor.{0,2} [256,256] */
    const float *critic0_w, *critic0_b, *critic2_w, *critic2_b;   /* base.critic.{0,2} [256,256] */
    const float *critic_linear_w, *critic_linear_b;               /* base.critic_linear [1,256] */
    const float *fc_mean_w, *fc_mean_b;                           /* dist.fc_mean [2,256] */
    const float *logstd;                                          /* dist.logstd._bias [2,1] */
} cn_policy_weights;

/* H humans, D = spatial edge width, max_envs = largest batch any later call will pass. */
int cn_policy_create(int human_num, int edge_width, int max_envs, cn_policy **out);
int cn_policy_destroy(cn_policy *p);
/* Snapshot the weights (copies + pre-folds the affine pairs q/k/v_linear∘in_proj and out_proj∘spatial_linear).
 * Call again after every optimiser step that changed them. */
int cn_policy_set_weights(cn_policy *p, const cn_policy_weights *w, void *stream);
/* One rollout-time forward for E envs.  hxs_in/out [E,128] (human_node_rnn), masks [E,1].  eps [E,2] standard-normal
 * noise or NULL for the deterministic mode() action.  Outputs value [E,1], action [E,2], logp [E,1]. */
int cn_policy_act(cn_policy *p, int E, const cn_obs *obs, const float *hxs_in, const float *masks, const float *eps,
                  float *value, float *action, float *logp, float *hxs_out, void *stream);
int cn_policy_get_value(cn_policy *p, int E, const cn_obs *obs, const float *hxs_in, const float *masks, float *value,
                        void *stream);
/* Test taps of the last act/get_value call: hh_out [E,H,512] is not materialised by the folded pipeline, so the taps
 * are spatial_lin [E,H,256], hr_attn [E,H], hr_out [E,256], robot_emb [E,256], actor_feat [E,256] (NULL to skip). */
int cn_policy_get_taps(cn_policy *p, int E, float *spatial_lin, float *hr_attn, float *hr_out, float *robot_emb,
                       float *actor_feat, void *stream);
/* Fused mode (cn_policy_set_gemm_mode 2): the robot-node kernel writes the taps above only while they are enabled (default 1);
 * rollout loops switch them off (17 MB of stores per 4096-env forward that nothing reads). */
int cn_policy_set_taps(cn_policy *p, int enabled);
/* fn(arg, stream) is called by cn_policy_act / cn_policy_get_value (fused mode) right after the human-human kernel was enqueued on `stream` and
 * before the robot-node kernel: the place for side work that must not reach the CUs before that kernel -- see cn_env_set_tail_deferral; pass
 * fn = (int (*)(void *, void *))cn_env_launch_tail and arg = the env batch.  A non-zero return aborts the forward with that status.  fn = NULL
 * removes the hook (do so before destroying what `arg` points to). */
int cn_policy_set_post_hh_hook(cn_policy *p, int (*fn)(void *arg, void *stream), void *arg);
/* Arithmetic of the three large human-human GEMMs (embedding_layer.2, folded q|k|v, folded out_proj∘spatial_linear):
 *   2 (default) = split precision (each fp32 operand as bf16 hi + lo, three bf16 MFMAs per term, fp32 accumulation; products
 *                 exact to ~2^-16 relative; outputs within 2e-5 of the fp32 path, bar 1e-4) with the whole human-human block as ONE
 *                 persistent kernel (activations never leave LDS / registers) and everything after it as ONE robot-node kernel;
 *   1           = the same split-precision arithmetic as separate launches (embedding / q|k|v / out_proj GEMMs, attention kernels,
 *                 per-env GEMMs);
 *   0           = exact fp32 on v_mfma_f32_32x32x2_f32, separate launches.
 * Everything outside the three large GEMMs always runs in exact fp32. */
int cn_policy_set_gemm_mode(cn_policy *p, int mode);
/* args.use_self_attn (arguments.py:189; selfAttn_srnn_temp_node.py:340-345, :402-414).  enabled = 1 (default): the human-human block is
 * SpatialEdgeSelfAttn + spatial_linear.  enabled = 0: no human-human attention -- spatial_linear is Sequential(Linear(D, 128), ReLU,
 * Linear(128, 256), ReLU) applied to the spatial edges themselves; cn_policy_set_weights then reads
 *   emb0_w / emb0_b               = base.spatial_linear.0 [128,D]
 *   spatial_linear_w / _b         = base.spatial_linear.2 [256,128]
 * and ignores emb2_*, q_*, k_*, v_*, in_proj_*, out_proj_* (they may be NULL).  Call before cn_policy_set_weights; changing it
 * invalidates the weight snapshot.  The robot-node part (robot-human attention, GRU, heads) is the same in both variants. */
int cn_policy_set_self_attention(cn_policy *p, int enabled);
/* args.sort_humans = False (arguments.py:206; selfAttn_srnn_temp_node.py:378-383, :402-414): the observation's humans are NOT sorted by
 * distance and both attention modules mask by `visible_masks` instead of by the detected count (all-invisible samples keep human 0).  Both
 * modules are permutation-equivariant over the humans and masked humans contribute exactly nothing, so the masked form equals the counted
 * form on the observation with the visible humans moved to the front (stable), which is what this entry point produces:
 *   spatial_edges [B,H,D], visible_masks [B,H] (bytes, non-zero = visible)  ->  out_edges [B,H,D] (visible rows first, the others behind
 *   them, each group in index order), out_detected [B] (float: max(1, number visible))
 * feed those to cn_policy_act / cn_hh_block_fwd / cn_rn_seq_fwd in place of the raw observation.  One wavefront per sample, H <= 64. */
int cn_obs_compact_visible(int B, int H, int D, const float *spatial_edges, const uint8_t *visible_masks, float *out_edges, float *out_detected,
                           void *stream);
/* Dominant-kernel timing support for bench.py: `every` = 0 switches it off, n >= 1 brackets every n-th forward's dominant kernel
 * (the fused human-human kernel, or the QKV projection of the separate-launch modes) with a pair of hipEvents on `stream`;
 * cn_policy_get_profile returns the accumulated device time [0], the bracketed launches [0] and their live rows [1].  An event
 * record costs a few microseconds of dispatch gap on the stream it sits on, hence the stride. */
int cn_policy_set_profiling(cn_policy *p, int every);
int cn_policy_get_profile(cn_policy *p, double *ms_out /*[8]*/, int64_t *launches_out /*[8]*/);
/* The same brackets one by one (bench.py reports their median / min / max): copies up to `cap` per-launch durations [ms], oldest first,
 * into ms_out (host memory) and returns how many there are (waits for the brackets still in flight); cn_policy_reset_profile drops
 * the samples and the sums but keeps the stride -- the events stay warm (a timing event's first record on a queue switches the
 * queue's profiling on, a one-off cost of some hundred microseconds that must not fall into a timed window). */
int cn_policy_get_profile_samples(cn_policy *p, float *ms_out, int cap);
int cn_policy_reset_profile(cn_policy *p);

/* ---- device-side launch stamps (measurement aid for bench.py / tools; not part of the reference's interface) ----
 * The kernels of the rollout step (CN_PROF_K_*) stamp the device's 100 MHz wall clock when their first workgroups start and when their
 * wavefronts end, into a caller-owned DEVICE ring of steps x CN_PROF_KERNELS slots of CN_PROF_SLOT_WORDS uint64.  A slot is 2 x 64
 * sub-slots of 16 words (one cache line each: same-address atomics serialise): word 16 b of the first half is a candidate for the
 * start (the minimum counts; initialise the slot's first half to all ones), word 16 (64 + b) one for the end (the maximum counts;
 * initialise to 0); word 1 of the slot carries a per-kernel count (the human-human kernel's live rows; initialise to 0).  Unlike an event
 * bracket on the stream the stamps cost no dispatch gap and do not include the time a launch waits for the host, and because the
 * clock is global the slots of a step also give its timeline.  cn_prof_set_stamps installs the ring for the whole process (NULL, 0
 * removes it) with a bit mask of the kernels to stamp; cn_prof_next_step moves to the next row (the first call selects row 0; rows
 * beyond `steps` are not stamped) and returns the row index.  Host-side state: call both from the thread that enqueues the step. */
enum { CN_PROF_K_ENV_STEP = 0, CN_PROF_K_ORCA_LANE = 1, CN_PROF_K_HH_FUSED = 2, CN_PROF_K_RN_FUSED = 3, CN_PROF_K_ORCA_LP3 = 4, CN_PROF_K_PREGEN = 5,
       CN_PROF_K_ROW_PLAN = 6, CN_PROF_K_OTHER = 7, CN_PROF_KERNELS = 8, CN_PROF_SLOT_WORDS = 2048 };
int cn_prof_set_stamps(uint64_t *ring, int steps, unsigned kernel_mask);
int cn_prof_next_step(void);

/* ---- the human-human block of evaluate_actions' forward as ONE launch (training path; crowds of <= 48 humans) ----
 * rl/networks/selfAttn_srnn_temp_node.py:63-91 + :408 on the compacted live rows, i.e. what cn_embed0_fwd -> cn_linear_fwd (embedding_layer.2,
 * ReLU) -> cn_linear_fwd (folded q|k|v) -> cn_hh_attention_fwd -> cn_linear_fwd (folded out_proj∘spatial_linear, ReLU) compute in five
 * launches with every activation passing through HBM twice; here the rollout's fused kernel runs on the training weights and writes each
 * activation the backward kernels need exactly once: e0 [R,128], x [R,512] (both post-ReLU), qkv [R,1536] (q unscaled; q_scale = 0.125
 * multiplies the scores), attn [R,512], out_sp [R,256] (post-ReLU).  spatial_edges [B,H,D] dense, row_off [B+1] (exclusive prefix of the
 * detected humans, R = row_off[B]); weights fp32 row-major ([512,128], [1536,512], [256,512]; biases 16-byte aligned); workspace =
 * cn_hh_block_workspace_bytes() bytes for the fragment-ordered bf16 hi/lo images, rebuilt on every call (the weights change every
 * optimiser step).  Same bf16x3 arithmetic as cn_linear_fwd.  The backward is the existing per-layer kernels on these outputs. */
int64_t cn_hh_block_workspace_bytes(void);
int cn_hh_block_fwd(int B, int H, int D, const float *spatial_edges, const int *row_off, const float *emb0_w, const float *emb0_b,
                    const float *emb2_w, const float *emb2_b, const float *qkv_w, const float *qkv_b, const float *os_w, const float *os_b,
                    float q_scale, void *workspace, float *e0, float *x, float *qkv, float *attn, float *out_sp, void *stream);

/* ---- the robot-node sequence of evaluate_actions' forward and backward as ONE call each (training path) ----
 * Everything behind the human-human block in rl/networks/model.py:82-90 -> selfAttn_srnn_temp_node.py:395-449 + srnn_model.py:35-105 +
 * distributions.py:36-44 for a [T, N] rollout slice (B = T * N samples, T-major): robot_linear -> [u | encoder_linear] -> robot-human attention over
 * the compacted out_sp rows -> edge_attention_embed -> GRU over the T steps with the done mask -> actor / critic trunks -> critic_linear and the
 * log-probability of the GIVEN actions under the DiagGaussian head.  Together with cn_hh_block_fwd (+ the per-layer backward kernels) this is
 * the train-mode policy forward of the boundary: no library GEMM and no framework pointwise kernel is left between the observation and
 * (value, log-prob).  Weights are the fp32 training weights as the host composes them (two affine pairs folded by the caller, whose autograd
 * carries the gradients of the folded matrices back to the factors): te = [spatial_edge_layer^T temporal_edge_layer ; encoder_linear],
 * ac0 = (actor.0 ; critic.0) o output_linear.  bf16x3 split-precision products (cn_linear_fwd_act) forward and for dX -- edge_attention_embed's
 * 64-output forward on the exact-fp32 kernel --, bf16x3 split-K for the weight gradients.
 * cn_rn_seq_fwd writes every activation the backward needs into the caller's `saved` buffers; cn_rn_seq_bwd takes d_value / d_logp [B] and
 * returns d_out_sp [R,256] (gradient into the human-human block), d_h0 [N,128] and the gradient of every weight (fixed summation orders). */
typedef struct {
    const float *rl_w, *rl_b;               /* robot_linear.0 [256,9], [256] */
    const float *te_w, *te_b;               /* [320,256], [320]: rows 0..255 u = Ws^T (Wt . + bt), rows 256..319 encoder_linear */
    const float *edge_w, *edge_b;           /* edge_attention_embed [64,256], [64] */
    const float *wih, *bih, *whh, *bhh;     /* GRU [384,128], [384] */
    const float *ac0_w, *ac0_b;             /* [512,128], [512]: (actor.0 ; critic.0) o output_linear */
    const float *a2_w, *a2_b, *c2_w, *c2_b; /* actor.2, critic.2 [256,256], [256] */
    const float *cl_w, *cl_b, *fm_w, *fm_b, *logstd; /* critic_linear [1,256],[1]; dist.fc_mean [2,256],[2]; dist.logstd [2] */
} cn_rn_weights;
typedef struct { /* gradients, same shapes as cn_rn_weights (device buffers, overwritten) */
    float *rl_w, *rl_b, *te_w, *te_b, *edge_w, *edge_b, *wih, *bih, *whh, *bhh, *ac0_w, *ac0_b, *a2_w, *a2_b, *c2_w, *c2_b, *cl_w, *cl_b, *fm_w, *fm_b, *logstd;
} cn_rn_grads;
typedef struct { /* activations kept for the backward (device buffers of the caller) */
    float *rs;    /* [B,256] relu(robot_linear) */
    float *z;     /* [B,384] u (256) | relu(enc) (64) | relu(edge) (64) */
    float *hr;    /* [B,256] attended human features */
    float *attn;  /* [B,H]   robot-human attention weights */
    float *gi;    /* [B,384] x W_ih^T + b_ih */
    float *hs;    /* [B,128] GRU outputs (hs[(T-1) N ..] is the final hidden state) */
    float *hms;   /* [B,128] masked previous states */
    float *gates; /* [B,512] r, z, n, gh_n */
    float *a1;    /* [B,512] tanh of the first trunk layers (actor | critic) */
    float *a2;    /* [B,512] tanh of the second trunk layers (actor | critic) */
} cn_rn_saved;
int64_t cn_rn_seq_workspace_floats(int T, int N);   /* scratch of cn_rn_seq_bwd */
int64_t cn_rn_seq_fwd_workspace_floats(void);       /* scratch of cn_rn_seq_fwd (the split planes of this step's weights) */
int cn_rn_seq_fwd(int T, int N, int H, const float *robot_node /*[B,7]*/, const float *temporal_edges /*[B,2]*/, const float *out_sp /*[R,256]*/,
                  const int *row_off /*[B+1]*/, const float *h0 /*[N,128]*/, const float *masks /*[B]*/, const float *actions /*[B,2]*/,
                  const cn_rn_weights *w, const cn_rn_saved *saved, float *workspace, float *value /*[B]*/, float *logp /*[B]*/, void *stream);
int cn_rn_seq_bwd(int T, int N, int H, const float *robot_node, const float *temporal_edges, const float *out_sp, const int *row_off, const float *masks,
                  const float *actions, const cn_rn_weights *w, const cn_rn_saved *saved, const float *d_value /*[B]*/, const float *d_logp /*[B]*/,
                  float *workspace, float *d_out_sp /*[R,256]*/, float *d_h0 /*[N,128]*/, const cn_rn_grads *grads, void *stream);

/* ---- human-human attention core, stand-alone (training path) ----
 * The (env, head) units of torch.nn.MultiheadAttention's scaled-dot-product core (selfAttn_srnn_temp_node.py:89) on COMPACTED
 * rows: sample b owns rows row_off[b] .. row_off[b+1]-1 (its detected humans); qkv [R,1536] = [q | k | v] (8 heads x 64).
 * fwd: out [R,512] = softmax(scale * q k^T) v per unit.  bwd: d_qkv [R,1536] from d_out [R,512] (softmax recomputed). */
/* cls: device workspace of cn_hh_attention_workspace_ints(B) int32 holding the size-class lists of the samples (<= 8 / 16 / 32 / 64 live
 * humans): each class is one launch that walks only its own (sample, head) units with an LDS footprint sized for it.  The forward builds
 * the lists when cls != NULL (NULL: every launch inspects every unit); the backward needs the buffer and rebuilds the lists unless
 * cls_ready != 0 (= the buffer was filled by the forward call on the same row_off). */
int64_t cn_hh_attention_workspace_ints(int B);
int cn_hh_attention_fwd(int B, int H, const float *qkv, const int *row_off, float scale, float *out, int *cls, void *stream);
int cn_hh_attention_bwd(int B, int H, const float *qkv, const int *row_off, const float *d_out, float scale, float *d_qkv, int *cls, int cls_ready,
                        void *stream);

/* ---- robot-human attention, stand-alone (training path) ----
 * EdgeAttention_M.att_func (rl/networks/selfAttn_srnn_temp_node.py:145-177) on COMPACTED rows: sample b owns rows
 * row_off[b] .. row_off[b+1]-1 of out_sp [R,256] (the attended values).  The reference scores t . s_j with
 * t = temporal_edge_layer(robot) and s_j = spatial_edge_layer(out_sp_j) = Ws out_sp_j + bs equal (Ws^T t) . out_sp_j up
 * to a per-sample constant the softmax ignores, so the op takes u [B,256] = Ws^T t instead of t and s.
 * fwd: attn [B,H] = softmax((H / 8) u . out_sp_j) (zero on padded humans), hr_out [B,256] = sum_j attn_j out_sp_j.
 * bwd: d_u [B,256] and d_o [R,256] from d_hr [B,256]. */
int cn_hr_attention_fwd(int B, int H, const float *u, const float *out_sp, const int *row_off, float *hr_out, float *attn,
                        void *stream);
int cn_hr_attention_bwd(int B, int H, const float *u, const float *out_sp, const int *row_off, const float *attn,
                        const float *d_hr, float *d_u, float *d_o, void *stream);

/* ---- embedding_layer.0 of the human-human block (training path) ----
 * Linear(D -> 128) + ReLU on the compacted rows (rl/networks/selfAttn_srnn_temp_node.py:33-36), D = 2 (VarNum) or 12
 * (Pred envs).  fwd: y [R,128] = relu(x [R,D] W^T + b).  bwd: dWb [128, D+1] = per output column the D weight
 * gradients followed by the bias gradient, from dy [R,128] masked by y > 0; `blocks` row-strided partial sums land in
 * partials [blocks,128,D+1] and are reduced in block order (deterministic).  The inputs are observations: no dx. */
int cn_embed0_fwd(int R, int D, const float *x, const float *W, const float *b, float *y, void *stream);
int cn_embed0_bwd(int R, int D, const float *x, const float *y, const float *dy, int blocks, float *partials, float *dWb,
                  void *stream);

/* ---- GRU cell of the human node RNN, pointwise part (training path) ----
 * torch.nn.GRU (gate order r,z,n) as EndRNN drives it one step at a time with h * done-mask
 * (rl/networks/srnn_model.py:35-105, selfAttn_srnn_temp_node.py:262-285), under autograd in PPO.update.
 * gi [N,384] = x W_ih^T + b_ih, gh [N,384] = hm W_hh^T + b_hh, hm [N,128] = masked previous hidden state.
 * fwd: h_out [N,128] and gates [N,512] = (r, z, n, gh_n) kept for the backward.
 * bwd: from dh [N,128]: dgi [N,384], dgh [N,384] and the direct path dhm [N,128] = dh * z. */
int cn_gru_cell_fwd(int N, const float *gi, const float *gh, const float *hm, float *h_out, float *gates, void *stream);
int cn_gru_cell_bwd(int N, const float *gates, const float *hm, const float *dh, float *dgi, float *dgh, float *dhm,
                    void *stream);
/* The whole sequence in one launch per direction (W_hh resident in registers, one workgroup per 32 rows):
 * gi [T,N,384] = x W_ih^T + b_ih, h0 [N,128], masks [T,N] -> hs [T,N,128]; hms [T,N,128] (masked previous states) and
 * gates [T,N,512] are kept for the backward.  bwd: d_hs [T,N,128] -> dgi [T,N,384], dgh [T,N,384] (the caller forms
 * d(W_hh) = dgh^T hms and d(b_hh) = column sums of dgh) and dh0 [N,128]. */
int cn_gru_seq_fwd(int T, int N, const float *gi, const float *h0, const float *masks, const float *w_hh, const float *b_hh,
                   float *hs, float *hms, float *gates, void *stream);
int cn_gru_seq_bwd(int T, int N, const float *gates, const float *hms, const float *masks, const float *w_hh,
                   const float *d_hs, float *dgi, float *dgh, float *dh0, void *stream);

/* ---- large Linear layers of the PPO update (training path), split-precision bf16x3 MFMA like the rollout forward ----
 * Replace torch.nn.Linear forward/backward of embedding_layer.2, the folded (q|k|v)_linear∘in_proj and the folded
 * out_proj∘spatial_linear (rl/networks/selfAttn_srnn_temp_node.py:63-91,408) as autograd runs them inside PPO.update
 * (rl/ppo.py:60-95).  All pointers are device pointers; hi/lo are bf16 planes (uint16 storage) of the weight.
 * cn_split_bf16:   w [rows,cols] fp32 -> hi, lo (bf16) of w (transpose = 0) or of w^T [cols,rows] (transpose = 1), stored in
 *                  the fragment order cn_linear_fwd loads (opaque: only cn_linear_fwd reads these planes).  The weight as
 *                  the product sees it must be [32 a, 16 b].
 * cn_linear_fwd:   Y[M,N] = act(X[M,K] W^T + bias), W given as hi/lo [N,K]; act 0 = none, 1 = ReLU; bias may be NULL.
 *                  With the transposed split of W [N,K] passed as a [K,N] weight it computes dX = dY W (no bias).
 *                  relu_gate (optional, same shape and leading dimension as X): X is replaced by X * [relu_gate > 0] while it
 *                  is loaded -- the backward of Linear+ReLU without a separate masking pass.  N % 128 == 0, K % 64 == 0.
 * cn_linear_fwd_act: the same product with the epilogues of the robot-node sequence (rl/networks/srnn_model.py actor / critic trunks: tanh): act 0 = none,
 *                  1 = ReLU, 2 = tanh, 3 = times relu'(.) = [aux > 0], 4 = times tanh'(.) = 1 - aux^2, aux [M, ldaux] being the forward VALUE of
 *                  the activation the product is a gradient of; columns >= relu_from get a ReLU on top (pass relu_from >= N for none).
 * cn_split_bf16_padded: cn_split_bf16 with the weight the product sees zero-padded to n_padded rows (a layer whose output count is not a
 *                  multiple of 128; 0 = no padding).
 * cn_linear_wgrad: dW[N,K] = dY[M,N]^T X[M,K] and (optional) db[N] = column sums of dY (dY gated by relu_gate > 0 when that
 *                  pointer, shaped like dY, is given).  The M reduction is cut into
 *                  `splits` ranges whose partial products land in partials [splits,N,K] (db_partials [splits,N]) and are
 *                  summed in split order (deterministic).  N % 64 == 0, K % 128 == 0.
 * cn_linear_wgrad_splits: the split count the library would pick for (M,N,K); 0 if the shape is unsupported. */
int cn_split_bf16(const float *w, int rows, int cols, int transpose, void *hi, void *lo, void *stream);
int cn_linear_fwd(int M, int N, int K, const float *X, int ldx, const float *relu_gate, const void *Whi, const void *Wlo,
                  const float *bias, int act, float *Y, int ldy, void *stream);
int cn_split_bf16_padded(const float *w, int rows, int cols, int transpose, int n_padded, void *hi, void *lo, void *stream);
int cn_linear_fwd_act(int M, int N, int K, const float *X, int ldx, const void *Whi, const void *Wlo, const float *bias, int act,
                      const float *aux, int ldaux, int relu_from, float *Y, int ldy, void *stream);
int cn_linear_wgrad_splits(int M, int N, int K);
/* C[M,N] (contiguous) = A . B in exact fp32, A and B addressed through element strides (A[m,k] = A[m * a_stride_m + k * a_stride_k], B[k,n] likewise):
 * the small weight-by-weight products of the update -- the affine folds of the mirror (policy.py: (q|k|v)_linear o in_proj, out_proj o
 * spatial_linear, Ws^T Wt, (actor.0 ; critic.0) o output_linear; selfAttn_srnn_temp_node.py:63-91, :160-163, :262-268 compute the unfolded
 * chains) and their backward, every transposed form being a choice of strides; N = 1 is a matrix-vector product.  Dimensions <= 4096. */
int cn_small_mm(int M, int N, int K, const float *A, int64_t a_stride_m, int64_t a_stride_k, const float *B, int64_t b_stride_k, int64_t b_stride_n,
                float *C, void *stream);
int cn_linear_wgrad(int M, int N, int K, const float *dY, int ldy, const float *relu_gate, const float *X, int ldx, int splits,
                    float *partials, float *db_partials, float *dW, float *db, void *stream);

/* ---- GST trajectory predictor + VecPretextNormalize (CrowdSimPredRealGST-v0, BASELINE configs[3]) ----
 * cn_gst_predict          <- gst_updated/scripts/wrapper/crowd_nav_interface_parallel.py:45-114 CrowdNavPredInterfaceMultiEnv.forward
 *                            (st_model.forward, gst_updated/src/gumbel_social_transformer/st_model.py:271-455, shipped hyper-parameters)
 * cn_gst_wrapper_reset    <- rl/vec_env/vec_pretext_normalize.py:85-101 VecPretextNormalize.reset (history buffers)
 * cn_gst_wrapper_step     <- rl/vec_env/vec_pretext_normalize.py:112-191 process_obs_rew
 * Device pointers to the fp32 parameters, named after the checkpoint's state_dict keys (epoch_100.pt). */
typedef struct {
    const float *node_embedding_w, *node_embedding_b;      /* gumbel_social_transformer.node_embedding [64,2] */
    const float *in_proj_w, *in_proj_b;                    /* ...node_encoder_layers.0.self_attn.in_proj_* [192,64] */
    const float *out_proj_w, *out_proj_b;                  /* ...self_attn.out_proj [64,64] */
    const float *norm_node_w, *norm_node_b;                /* ...norm_node [64] */
    const float *norm1_node_w, *norm1_node_b;              /* ...norm1_node [64] */
    const float *linear1_w, *linear1_b;                    /* ...linear1 [128,64] */
    const float *linear2_w, *linear2_b;                    /* ...linear2 [64,128] */
    const float *lstm_w_ih, *lstm_w_hh, *lstm_b_ih, *lstm_b_hh; /* lstm.*_l0 [256,64],[256,64],[256],[256] */
    const float *hidden2pos_w, *hidden2pos_b;              /* hidden2pos [5,64] */
} cn_gst_weights;
typedef struct cn_gst cn_gst;
int cn_gst_create(int human_num, int max_envs, cn_gst **out);
int cn_gst_destroy(cn_gst *g);
int cn_gst_set_weights(cn_gst *g, const cn_gst_weights *w, void *stream);
/* in_traj [E,H,5,2] world positions, in_mask [E,H,5] (0/1 float) -> out_traj [E,H,5,5] = cumulative (mu_x, mu_y, sigma_x,
 * sigma_y, corr), positions -999 where the pedestrian is not predicted; out_mask [E,H] (0/1 float). */
int cn_gst_predict(cn_gst *g, int E, const float *in_traj, const float *in_mask, float *out_traj, float *out_mask, void *stream);
int cn_gst_wrapper_reset(cn_gst *g, int E, void *stream);
/* Prediction stride (rl/vec_env/vec_pretext_normalize.py:56-57, :133-134): pred_interval = int(data.pred_timestep // env.time_step); the
 * wrapper keeps the last (5 - 1) * pred_interval + 1 observations and feeds every pred_interval-th of them (oldest first) to the predictor.
 * Default 1 (every shipped config).  Re-allocates the history: call before cn_gst_wrapper_reset.  cn_gst_wrapper_history_len = that length. */
int cn_gst_wrapper_set_interval(cn_gst *g, int pred_interval);
int cn_gst_wrapper_history_len(const cn_gst *g);
/* Checkpointing of the wrapper's observation history (traj_buffer / mask_buffer, vec_pretext_normalize.py:85-101), in time order, oldest first:
 * traj [len,E,H,2] float32, mask [len,E,H] uint8 (device buffers), len = cn_gst_wrapper_history_len.  With the simulator snapshot
 * (cn_env_save) this makes a resumed CrowdSimPredRealGST-v0 run continue bit for bit. */
int cn_gst_wrapper_save(cn_gst *g, float *traj, uint8_t *mask, void *stream);
int cn_gst_wrapper_load(cn_gst *g, int E, const float *traj, const uint8_t *mask, void *stream);
/* obs: the raw CrowdSimPredRealGST-v0 observation (robot_node, spatial_edges [E,H,12] by human id, visible_masks).  Pushes the
 * new positions into the 5-deep history, runs the predictor, adds the social penalty min_{h,k}(collision * penalty / 2^(k+2))
 * to rewards [E] (in place, may be NULL) and writes spatial_edges_out [E,H,12]: predictions in the robot frame where valid,
 * rows sorted by current distance. */
int cn_gst_wrapper_step(cn_gst *g, int E, const cn_obs *obs, float robot_plus_human_radius, float collision_penalty, float *rewards,
                        float *spatial_edges_out, void *stream);

/* ---- GST predictor TRAINING step: forward + negative log-likelihood + backward of one batch of sequences ----
 * gst_updated/scripts/experiments/train.py:107-146 (loop body) over st_model.py:271-455 (training-time forward: 'faster_lstm', recursive decoding on
 * the mean, sampling = False) and :62-112 (negative_log_likelihood_full_partial), shipped hyper-parameters.  One workgroup per sequence:
 *   v_obs [B,5,N,2], v_pred [B,5,N,2]: displacements of the N pedestrians (seq_to_graph's vertices; -999 where missing),
 *   loss_mask_rel [B,N,10]: 1 where the displacement exists (the per-step attention masks are its outer products, trajectories.py:131-133).
 * 4 <= N <= 64 (pad a smaller crowd with absent pedestrians: mask rows of zeros).  grads: same struct as the weights, every field WRITTEN with
 * d(loss)/d(parameter), loss = sum of the masked NLL over the batch / number of valid (step, pedestrian) pairs (train.py:131-133; B = 1 is
 * the shipped batch size).  p_drop: the reference's dropout (0.1 while training; four sites), masks from a counter-based hash of `seed` --
 * this library's own stream, not torch's; p_drop = 0 reproduces the reference's gradients.  loss_out [2] = loss, valid-pair count;
 * gauss_out (optional) [B,5,N,5] = mu_x, mu_y, sigma_x, sigma_y, corr of every predicted step.  The optimiser step is cn_adam_clip_step. */
int64_t cn_gst_train_workspace_bytes(int B, int N);
int cn_gst_train_step(int B, int N, const float *v_obs, const float *v_pred, const float *loss_mask_rel, const cn_gst_weights *w, const cn_gst_weights *grads,
                      float p_drop, uint64_t seed, void *workspace, int64_t workspace_bytes, float *loss_out, float *gauss_out, void *stream);

/* ---- rollout math ---- */
/* rewards [T,N], values [T+1,N], masks [T+1,N] -> returns[t][n] for t < T (row T untouched).  fp32, torch op order. */
int cn_gae(int T, int N, const float *rewards, const float *values, const float *masks, double gamma, double lam,
           float *returns, void *stream);
/* stats[3] (float64, device) = {sum, sum of squares, count} of (returns - values) over n elements */
int cn_adv_stats(int64_t n, const float *returns, const float *values, double *stats, void *stream);
/* adv = ((returns - values) - mean) / (std_unbiased + 1e-5) with mean/std derived from stats (possibly all-reduced) */
int cn_adv_normalize(int64_t n, const float *returns, const float *values, const double *stats, float *adv, void *stream);

/* bench.Monitor's aggregate of one vec-env step (rl/networks/envs.py:70-73, train.py:180-182) from the outputs of cn_env_step: acc[8] (float64,
 * device) += {finished episodes, sum of their returns, sum of their lengths, timeouts, collisions, goals reached, -, -}.  One launch with a
 * fixed summation order instead of two dozen small reductions per rollout step. */
int cn_episode_stats_update(int E, const uint8_t *done, const uint8_t *info, const double *ep_return, const int32_t *ep_len, double *acc, void *stream);

/* ---- PPO losses (rl/ppo/ppo.py:66-84) ----
 * values, logp (new policy), old_logp, adv (normalised advantages), value_preds, returns: [n] float32 (the [T*N,1] minibatch
 * tensors of recurrent_generator).  fwd: losses[0] = value_loss = 0.5 * mean(max((v - R)^2, (vp + clamp(v - vp, +-clip) - R)^2))
 * (or 0.5 * mean((R - v)^2) when use_clipped_value_loss == 0), losses[1] = action_loss = -mean(min(ratio * A,
 * clamp(ratio, 1 - clip, 1 + clip) * A)), ratio = exp(logp - old_logp).  workspace: cn_ppo_loss_workspace_doubles() float64.
 * bwd: d_values[n], d_logp[n] = gradients of g_losses[0] * value_loss + g_losses[1] * action_loss (g_losses: 2 floats on the
 * DEVICE, the upstream gradients autograd hands in -- value_loss_coef and 1 in ppo.py:86); torch's tie rules for min / max /
 * clamp.  The entropy term of ppo.py:86 depends on dist.logstd only and stays in the host mirror. */
int cn_ppo_loss_workspace_doubles(void);
int cn_ppo_loss_fwd(int64_t n, const float *values, const float *logp, const float *old_logp, const float *adv,
                    const float *value_preds, const float *returns, float clip_param, int use_clipped_value_loss,
                    double *workspace, float *losses, void *stream);
int cn_ppo_loss_bwd(int64_t n, const float *values, const float *logp, const float *old_logp, const float *adv,
                    const float *value_preds, const float *returns, float clip_param, int use_clipped_value_loss,
                    const float *g_losses, float *d_values, float *d_logp, void *stream);

/* ---- gradient-norm clip + Adam over one flat bucket (rl/ppo/ppo.py:88-90, optimiser of ppo.py:32) ----
 * param, grad, exp_avg, exp_avg_sq: [n] float32, the concatenation of all parameters / their gradients / Adam moments in
 * Module.parameters() order (the host mirror makes every p.data / p.grad / optimizer.state[p] a view into these).
 * grad <- grad * grad_scale (1 / world_size after a sum all-reduce, else 1); total_norm = ||grad||_2;
 * grad <- grad * min(1, max_grad_norm / (total_norm + 1e-6)) (skipped when max_grad_norm <= 0); then torch.optim.Adam's
 * update for step number `step` (1-based), no weight decay / amsgrad.  workspace: cn_adam_workspace_doubles() float64;
 * grad_norm_out (optional, 1 float, device) receives total_norm.  Two launches, no host synchronisation. */
int cn_adam_workspace_doubles(void);
int cn_adam_clip_step(int64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, double grad_scale,
                      double max_grad_norm, double lr, double beta1, double beta2, double eps, int64_t step, double *workspace,
                      float *grad_norm_out, void *stream);

/* ---- one PPO minibatch step as ONE boundary call (rl/ppo/ppo.py:55-95 for one `sample` of the recurrent generator) ----
 * Replaces, for the default network (use_self_attn, sort_humans; crowds of <= 48 humans, edge width <= 16), the statement sequence
 *   sample = recurrent_generator(...)            rl/networks/storage.py:184-253  (gather of N whole env trajectories by index)
 *   evaluate_actions(...)                        rl/networks/model.py:82-90
 *   value_loss / action_loss / entropy           rl/ppo/ppo.py:66-86
 *   optimizer.zero_grad(); total.backward()      rl/ppo/ppo.py:87-88
 * by: gather -> affine folds -> cn_hh_block_fwd -> cn_rn_seq_fwd -> cn_ppo_loss_fwd/bwd -> cn_rn_seq_bwd -> the per-layer backward of
 * the human-human block -> chain rule of the folds, all on `stream` (+ a library-owned side stream that carries weight-gradient
 * products off the critical path and is joined before the call returns its last launch), with NO framework kernel in between: every
 * parameter gradient is WRITTEN (not accumulated) to grads-><same field>; parameters the loss does not reach
 * (attn.spatial_edge_layer.bias) get an exact zero.  The caller then runs its gradient all-reduce (data parallel) and
 * cn_adam_clip_step on the flat bucket the pointers live in.
 *
 * storage tensors (float32, contiguous, device; T = num_steps, E = envs in storage):
 *   robot_node [T+1,E,1,7], temporal_edges [T+1,E,1,2], spatial_edges [T+1,E,H,D], detected_human_num [T+1,E,1], h0 = recurrent_hidden_states
 *   ['human_node_rnn'][0] [E,1,128], masks [T+1,E,1], actions [T,E,2], value_preds / returns [T+1,E,1], old_logp = action_log_probs [T,E,1],
 *   adv = normalised advantages [T,E,1];  env_idx [N] int32 = perm[start : start + N] (storage.py:209-210).
 * rows = sum over the minibatch of clamp(detected_human_num, 1, H) -- the caller knows it from cn_ppo_row_totals (one readback per
 * update(), not per minibatch); the call fails with CN_ERR_INVALID if the workspace is smaller than cn_ppo_minibatch_workspace_bytes.
 * losses_out [3] (device) = value_loss, action_loss, dist_entropy (ppo.py:66-86, the three numbers update() averages). */
typedef struct {
    int T, N, E, H, D;
    const int32_t *env_idx;
    const float *robot_node, *temporal_edges, *spatial_edges, *detected_human_num, *h0, *masks, *actions, *value_preds, *returns, *old_logp, *adv;
} cn_ppo_batch;
typedef struct {
    float clip_param, value_loss_coef, entropy_coef;
    int use_clipped_value_loss;
} cn_ppo_hyper;
int64_t cn_ppo_minibatch_workspace_bytes(int T, int N, int H, int D, int64_t rows);
/* totals [E] int32 (device) = sum_t clamp(detected_human_num[t, e], 1, H) over t < T: the compacted rows env e contributes to a minibatch */
int cn_ppo_row_totals(int T, int E, int H, const float *detected_human_num, int32_t *totals, void *stream);
int cn_ppo_minibatch_step(const cn_ppo_batch *batch, int64_t rows, const cn_policy_weights *params, const cn_policy_weights *grads,
                          const cn_ppo_hyper *hyper, void *workspace, int64_t workspace_bytes, float *losses_out,
                          float *value_logp_out /* optional [2, T * N]: values, log-probs of the minibatch (tests) */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CROWDNAV_HIP_H */
