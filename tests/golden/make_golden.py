#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the *Python reference* (build container only).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixtures are data only (inputs + the reference's outputs).  The reference itself never ships.
See _ref_import.py for the import recipe and the rvo2 caveat (ORCA arithmetic comes from the oracle's
RVO2 restatement because rvo2 is not available; everything around it is the reference's own Python).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402

_ARGV = sys.argv[1:]
R.install()

INFO_CODE = {"Nothing": 0, "Timeout": 1, "Collision": 2, "ReachGoal": 3, "Danger": 4}


def info_code(obj):
    return INFO_CODE[type(obj).__name__]


def make_env(env_name, cfg, seed, rank, nenv):
    import crowd_sim.envs as E

    cls = {"CrowdSimVarNum-v0": E.CrowdSimVarNum, "CrowdSimPred-v0": E.CrowdSimPred,
           "CrowdSimPredRealGST-v0": E.CrowdSimPredRealGST}[env_name]
    env = cls()
    env.configure(cfg)
    # rl/networks/envs.py:49-58
    env.thisSeed = seed + rank
    env.nenv = nenv
    env.phase = "train" if nenv > 1 else "test"
    return env


def cast_obs(ob, H):
    """What the vec-env does: copy into float32 / bool shared buffers (shmem_vec_env.py:124-129)."""
    out = {
        "robot_node": np.asarray(ob["robot_node"], dtype=np.float32).reshape(1, 7),
        "temporal_edges": np.asarray(ob["temporal_edges"], dtype=np.float32).reshape(1, 2),
        "spatial_edges": np.asarray(ob["spatial_edges"], dtype=np.float32),
        "detected_human_num": np.asarray([ob["detected_human_num"]], dtype=np.float32),
    }
    if "visible_masks" in ob:
        out["visible_masks"] = np.asarray(ob["visible_masks"], dtype=bool)
    else:
        out["visible_masks"] = np.zeros(H, dtype=bool)
    return out


def widen_unicycle_actions(env):
    """numpy promotion shim for the unicycle path.  The reference pins numpy 1.20.3, where a float32 scalar combined with a
    Python float / int gives float64 (`self.desiredVelocity[0] + action.v`, `self.theta + action.r`, `action.r ** 2`, ...), so
    everything after srnn.clip_action runs in float64.  This container only has numpy 2 (NEP 50: the same expressions stay
    float32).  The reference's own clip_action still does the clipping (float32, identical under both numpys); its two results are
    re-typed as np.float64 -- same values -- which makes every later expression promote exactly as under 1.20.3."""
    from crowd_sim.envs.utils.action import ActionRot
    inner = env.robot.policy.clip_action

    def clip_action(raw_action, v_pref):
        a = inner(raw_action, v_pref)
        return ActionRot(np.float64(a.v), np.float64(a.r))
    env.robot.policy.clip_action = clip_action


def unicycle_action_for(env, t, ep, rank):
    """(change of speed, change of heading) script: steer at the goal with gains that exercise both clips, some exact zeros
    (the R = 0 branch of the differential drive), a reversing phase at the start of some episodes."""
    err = np.arctan2(env.robot.gy - env.robot.py, env.robot.gx - env.robot.px) - env.robot.theta
    err = (err + np.pi) % (2 * np.pi) - np.pi
    mode = (ep + rank) % 3
    if mode == 0:
        a = np.array([0.2, 0.9 * err])
    elif mode == 1:
        a = np.array([-0.15 if t % 50 < 8 else 0.05 + 0.04 * np.sin(0.3 * t), 0.05 * err + 0.02 * np.sin(0.21 * t)])
    else:
        a = np.array([0.03, 0.0 if t % 7 == 0 else 0.3 * err])
    return a.astype(np.float32)


def action_for(env, t, ep, rank):
    """Deterministic action script (never touches the global numpy RNG the env uses)."""
    if env.robot.kinematics == "unicycle":
        return unicycle_action_for(env, t, ep, rank)
    gx, gy = env.robot.gx - env.robot.px, env.robot.gy - env.robot.py
    n = max(np.hypot(gx, gy), 1e-9)
    mode = (ep + rank) % 4
    if mode == 0:    # goal seeking, over-speed (exercises clip_action)
        a = np.array([1.7 * gx / n, 1.7 * gy / n])
    elif mode == 1:  # goal seeking with a wobble
        a = np.array([0.9 * gx / n + 0.5 * np.sin(0.37 * t), 0.9 * gy / n + 0.5 * np.cos(0.23 * t)])
    elif mode == 2:  # slow drift -> timeouts / danger
        a = np.array([0.12 * np.sin(0.11 * t + rank), 0.12 * np.cos(0.07 * t)])
    else:            # head for the centre then the goal
        a = np.array([-0.8 * env.robot.px / 6.0 + 0.5 * gx / n, -0.8 * env.robot.py / 6.0 + 0.5 * gy / n])
    return a.astype(np.float32)


def trace(env_name, over, seed, rank, nenv, steps, tag, phase=None):
    cfg = R.make_config(**over)
    cfg.args.env_name = env_name
    env = make_env(env_name, cfg, seed, rank, nenv)
    if phase is not None:   # make_env follows rl/networks/envs.py (train / test by env count); 'val' is only reachable by setting it
        env.phase = phase
    if cfg.action_space.kinematics == "unicycle":
        widen_unicycle_actions(env)
    H = cfg.sim.human_num + cfg.sim.human_num_range   # rows of every observation; len(env.humans) may be smaller

    def padded(rows, width):
        out = np.full((H, width), np.nan)
        if len(rows):
            out[:len(rows)] = np.asarray(rows, dtype=np.float64)
        return out
    rec = {k: [] for k in ("actions", "reward", "done", "info", "ep_return", "ep_len", "robot_state", "human_state",
                           "robot_node", "temporal_edges", "spatial_edges", "detected_human_num", "visible_masks",
                           "human_action", "min_dist", "human_count")}
    ob0 = cast_obs(env.reset(), H)
    init_humans = np.array([[h.px, h.py, h.vx, h.vy, h.gx, h.gy, h.radius, h.v_pref] for h in env.humans])
    init_robot = np.array(env.robot.get_full_state_list(), dtype=np.float64)
    ep, ep_ret, ep_len = 0, 0.0, 0
    rets = []
    for t in range(steps):
        a = action_for(env, t, ep, rank)
        rec["actions"].append(a.copy())
        pre = np.array([[h.px, h.py] for h in env.humans])
        ob, reward, done, info = env.step(a.copy())
        # human actions = displacement / dt is lossy; read velocities of humans that were not respawned instead
        rec["human_action"].append(padded([[h.vx, h.vy] for h in env.humans], 2))
        rets.append(reward)
        ep_len += 1
        code = info_code(info["info"])
        rec["min_dist"].append(float(getattr(info["info"], "min_dist", 0.0)) if code == 4 else 0.0)
        if done:
            ep_ret = round(sum(rets), 6)
            rec["ep_return"].append(ep_ret)
            rec["ep_len"].append(ep_len)
            ob = env.reset()
            rets, ep_len = [], 0
            ep += 1
        else:
            rec["ep_return"].append(np.nan)
            rec["ep_len"].append(0)
        ob = cast_obs(ob, H)
        for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num", "visible_masks"):
            rec[k].append(ob[k])
        rec["reward"].append(np.float32(reward))
        rec["done"].append(bool(done))
        rec["info"].append(code)
        rec["robot_state"].append(np.array(env.robot.get_full_state_list(), dtype=np.float64))
        rec["human_state"].append(padded([[h.px, h.py, h.gx, h.gy, h.radius, h.v_pref] for h in env.humans], 6))
        rec["human_count"].append(len(env.humans))
    out = {k: np.array(v) for k, v in rec.items()}
    out["reward64"] = np.array([float(x) for x in rec["reward"]])
    for k, v in ob0.items():
        out["reset_" + k] = v
    out["init_humans"] = init_humans
    out["init_robot"] = init_robot
    meta = dict(env_name=env_name, over=over, seed=seed, rank=rank, nenv=nenv, steps=steps, sort_humans=bool(cfg.args.sort_humans))
    if phase is not None:
        meta["phase"] = phase
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, "env_%s.npz" % tag)
    np.savez_compressed(path, **out)
    dn = int(np.sum(out["done"]))
    print("%-28s steps=%d episodes=%d infos=%s  -> %s (%.0f KB)" % (
        tag, steps, dn, np.bincount(out["info"], minlength=5).tolist(), os.path.basename(path), os.path.getsize(path) / 1024))


NON_RAND = {"env.randomize_attributes": False, "humans.random_goal_changing": False, "humans.end_goal_changing": True,
            "sim.predict_method": "none", "env.use_wrapper": False}
RAND = {"env.randomize_attributes": True, "humans.random_goal_changing": True, "humans.end_goal_changing": True,
        "sim.predict_method": "none", "env.use_wrapper": False}


def env_goldens():
    for rank in (0, 1):
        trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 20}), 425, rank, 4, 320, "varnum_h20_nonrand_r%d" % rank)
    for rank in (0, 3):
        trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 5}), 425, rank, 4, 420, "varnum_h5_rand_r%d" % rank)
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 50}), 425, 2, 8, 130, "varnum_h50_rand_r2")
    for rank in (0, 1):
        trace("CrowdSimPred-v0", dict(NON_RAND, **{"sim.human_num": 20, "sim.predict_method": "const_vel"}), 425, rank, 4,
              260, "pred_h20_constvel_r%d" % rank)
    trace("CrowdSimPred-v0", dict(RAND, **{"sim.human_num": 10, "sim.predict_method": "const_vel"}), 77, 1, 2, 200,
          "pred_h10_rand_r1")
    trace("CrowdSimPredRealGST-v0", dict(NON_RAND, **{"sim.human_num": 20, "sim.predict_method": "inferred"}), 425, 0, 4,
          140, "predgst_h20_r0")
    # nenv = 1 -> phase 'test' (rl/networks/envs.py:55-58): seeds 1000 + case, 'truth' roll-out every step, Danger decided by
    # the humans' true future positions (min_dist recorded), CrowdSimPred's social reward on the true futures
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 20}), 425, 0, 1, 260, "varnum_h20_test_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 5}), 425, 0, 1, 300, "varnum_h5_rand_test_r0")
    trace("CrowdSimPred-v0", dict(NON_RAND, **{"sim.human_num": 20, "sim.predict_method": "const_vel"}), 425, 0, 1, 200,
          "pred_h20_constvel_test_r0")
    # robot.policy = 'orca' (trained_models/ORCA_no_rand): the robot's action is ORCA on its beliefs, the passed action is ignored
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 20, "robot.policy": "orca"}), 425, 0, 1, 400, "varnum_h20_orcarobot_test_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 10, "robot.policy": "orca"}), 425, 0, 1, 300, "varnum_h10_rand_orcarobot_test_r0")
    # humans.policy = 'social_force': no rvo2 anywhere -> these traces are the reference's own arithmetic end to end (oracle only)
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 20, "humans.policy": "social_force"}), 425, 0, 4, 300, "varnum_h20_sfhumans_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 10, "humans.policy": "social_force", "robot.visible": True}), 425, 1, 4, 300,
          "varnum_h10_rand_sfhumans_robotvisible_r1")
    # robot.visible = True: every human's ORCA gets the robot as one more neighbour (crowd_sim.py:695-699)
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 20, "robot.visible": True}), 425, 1, 4, 300, "varnum_h20_robotvisible_r1")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 7, "robot.visible": True}), 425, 2, 4, 320, "varnum_h7_rand_robotvisible_r2")
    # sim.human_num_range > 0: the crowd size is drawn at reset and humans leave / arrive every 5 s (oracle only so far)
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 5, "sim.human_num_range": 2}), 425, 0, 4, 600, "varnum_h5_range2_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 20, "sim.human_num_range": 5}), 425, 1, 4, 400, "varnum_h20_rand_range5_r1")
    trace("CrowdSimPred-v0", dict(NON_RAND, **{"sim.human_num": 10, "sim.human_num_range": 3, "sim.predict_method": "const_vel"}), 425, 0, 4,
          300, "pred_h10_range3_r0")
    trace("CrowdSimPredRealGST-v0", dict(RAND, **{"sim.human_num": 12, "sim.human_num_range": 4, "sim.predict_method": "inferred"}), 425, 2, 4,
          240, "predgst_h12_rand_range4_r2")
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 15, "sim.human_num_range": 5}), 425, 0, 1, 300, "varnum_h15_range5_test_r0")
    # sim.predict_method = 'truth' as the OBSERVATION predictor of CrowdSimPred-v0 (oracle only so far)
    trace("CrowdSimPred-v0", dict(NON_RAND, **{"sim.human_num": 20, "sim.predict_method": "truth"}), 425, 1, 4, 260, "pred_h20_truthobs_r1")
    trace("CrowdSimPred-v0", dict(RAND, **{"sim.human_num": 8, "sim.human_num_range": 2, "sim.predict_method": "truth"}), 425, 0, 4, 300,
          "pred_h8_rand_range2_truthobs_r0")
    trace("CrowdSimPred-v0", dict(RAND, **{"sim.human_num": 10, "sim.predict_method": "truth"}), 425, 0, 1, 200, "pred_h10_rand_truthobs_test_r0")
    # action_space.kinematics = 'unicycle' (CrowdSimVarNum-v0; oracle only so far).  The reset draws 1 .. human_num + range humans
    # and the step asserts human_num - range <= len(humans) (:439), so the reference only runs with range = human_num - 1.
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 3, "sim.human_num_range": 2, "action_space.kinematics": "unicycle"}),
          425, 0, 4, 700, "varnum_h3_unicycle_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 6, "sim.human_num_range": 5, "action_space.kinematics": "unicycle"}),
          425, 1, 4, 500, "varnum_h6_rand_unicycle_r1")

    # ---- round 3: settings that used to raise NotImplementedError --------------------------------------------------------------
    # robot.visible in the TEST phase: the 'truth' roll-outs pass each human its H - 1 fellow humans only (crowd_sim_var_num.py:183-190),
    # the real step passes them plus the robot (crowd_sim.py:695-699) -> every private rvo2 simulator is rebuilt twice per step
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 10, "robot.visible": True}), 425, 0, 1, 300, "varnum_h10_robotvisible_test_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 8, "robot.visible": True}), 425, 0, 1, 300, "varnum_h8_rand_robotvisible_test_r0")
    # ... and with 'truth' as the observation predictor of CrowdSimPred-v0 (train and test phase)
    trace("CrowdSimPred-v0", dict(NON_RAND, **{"sim.human_num": 10, "robot.visible": True, "sim.predict_method": "truth"}), 425, 1, 4, 260,
          "pred_h10_truthobs_robotvisible_r1")
    trace("CrowdSimPred-v0", dict(RAND, **{"sim.human_num": 7, "robot.visible": True, "sim.predict_method": "truth"}), 425, 0, 1, 240,
          "pred_h7_rand_truthobs_robotvisible_test_r0")
    trace("CrowdSimPredRealGST-v0", dict(NON_RAND, **{"sim.human_num": 8, "robot.visible": True, "sim.predict_method": "inferred"}), 425, 1, 4, 200,
          "predgst_h8_robotvisible_r1")
    trace("CrowdSimPredRealGST-v0", dict(RAND, **{"sim.human_num": 8, "robot.visible": True, "sim.predict_method": "inferred"}), 425, 0, 1, 200,
          "predgst_h8_rand_robotvisible_test_r0")
    # humans.policy = 'social_force' with 'truth' roll-outs (test phase; observation predictor): the roll-outs roll the humans' OWN policy
    # (act_joint_state -> SOCIAL_FORCE.predict on the rolled states, crowd_sim_var_num.py:180-198) -- no rvo2 anywhere in these traces
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 10, "humans.policy": "social_force"}), 425, 0, 1, 300, "varnum_h10_sfhumans_test_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 8, "humans.policy": "social_force", "robot.visible": True}), 425, 0, 1, 300,
          "varnum_h8_rand_sfhumans_robotvisible_test_r0")
    trace("CrowdSimPred-v0", dict(NON_RAND, **{"sim.human_num": 10, "humans.policy": "social_force", "sim.predict_method": "truth"}), 425, 1, 4, 260,
          "pred_h10_sfhumans_truthobs_r1")
    trace("CrowdSimPred-v0", dict(RAND, **{"sim.human_num": 9, "sim.human_num_range": 3, "humans.policy": "social_force", "sim.predict_method": "const_vel"}),
          425, 0, 1, 260, "pred_h9_rand_range3_sfhumans_test_r0")
    trace("CrowdSimPredRealGST-v0", dict(NON_RAND, **{"sim.human_num": 8, "humans.policy": "social_force", "sim.predict_method": "inferred"}), 425, 0, 1, 200,
          "predgst_h8_sfhumans_test_r0")
    # action_space.kinematics = 'unicycle' in CrowdSimPred-v0 / CrowdSimPredRealGST-v0: CrowdSimPred.step sends the command through
    # smooth_action (the Turtlebot wheel model with np.random.normal dead-band noise; low-pass filtered in the test phase) and always
    # respawns the humans that reached their goal (crowd_sim_pred.py:120-131, :208-212)
    trace("CrowdSimPred-v0", dict(NON_RAND, **{"sim.human_num": 3, "sim.human_num_range": 2, "action_space.kinematics": "unicycle",
                                               "sim.predict_method": "const_vel"}), 425, 0, 4, 700, "pred_h3_unicycle_r0")
    trace("CrowdSimPred-v0", dict(RAND, **{"sim.human_num": 6, "sim.human_num_range": 5, "action_space.kinematics": "unicycle",
                                           "sim.predict_method": "const_vel"}), 425, 0, 1, 500, "pred_h6_rand_unicycle_test_r0")
    trace("CrowdSimPredRealGST-v0", dict(NON_RAND, **{"sim.human_num": 4, "sim.human_num_range": 3, "action_space.kinematics": "unicycle",
                                                      "sim.predict_method": "inferred"}), 425, 1, 4, 500, "predgst_h4_unicycle_r1")
    # phase 'val' (CrowdSimPred-v0; the other env classes fail there): seeds 0 + case, case counter modulo env.val_size, Danger decided by
    # the predictions the PREVIOUS observation left in self.human_future_traj
    trace("CrowdSimPred-v0", dict(NON_RAND, **{"sim.human_num": 12, "sim.predict_method": "const_vel"}), 425, 1, 4, 300, "pred_h12_constvel_val_r1", phase="val")
    trace("CrowdSimPred-v0", dict(RAND, **{"sim.human_num": 8, "sim.human_num_range": 2, "sim.predict_method": "truth", "env.val_size": 3}), 425, 0, 2, 300,
          "pred_h8_rand_range2_truthobs_val_r0", phase="val")
    # robot.FOV / humans.FOV below 2 (x pi): the robot only detects humans inside the cone around its heading (the direction of its velocity
    # for a holonomic robot), a human's ORCA / social force gets the dummy at (7, 7) for every agent outside its own cone (crowd_sim.py:513-552)
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 10, "robot.FOV": 1.0}), 425, 0, 4, 300, "varnum_h10_robotfov_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 8, "humans.FOV": 1.2, "robot.visible": True}), 425, 1, 4, 300, "varnum_h8_rand_humanfov_robotvisible_r1")
    trace("CrowdSimPred-v0", dict(NON_RAND, **{"sim.human_num": 10, "robot.FOV": 0.8, "humans.FOV": 1.5, "sim.predict_method": "const_vel"}), 425, 0, 1, 260,
          "pred_h10_fov_test_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 6, "humans.policy": "social_force", "humans.FOV": 1.0, "robot.FOV": 1.5}), 425, 0, 4, 300,
          "varnum_h6_rand_sfhumans_fov_r0")
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **{"sim.human_num": 3, "sim.human_num_range": 2, "action_space.kinematics": "unicycle", "robot.FOV": 1.0,
                                                 "humans.FOV": 1.0}), 425, 0, 4, 500, "varnum_h3_unicycle_fov_r0")
    trace("CrowdSimVarNum-v0", dict(RAND, **{"sim.human_num": 8, "robot.FOV": 1.0, "humans.FOV": 1.0, "robot.visible": True}), 425, 0, 1, 300,
          "varnum_h8_rand_fov_robotvisible_test_r0")

    # ---- round 4: data.pred_timestep = 2 x env.time_step (pred_interval 2, crowd_sim.py:180-181): const_vel predictions lie 0.5 s apart
    # (crowd_sim_var_num.py:212), 'truth' rolls the humans predict_steps * 2 times and keeps every second state (:181, :206)
    S2 = {"data.pred_timestep": 0.5}
    trace("CrowdSimPred-v0", dict(NON_RAND, **dict(S2, **{"sim.human_num": 12, "sim.predict_method": "const_vel"})), 425, 1, 4, 260, "pred_h12_constvel_stride2_r1")
    trace("CrowdSimPred-v0", dict(RAND, **dict(S2, **{"sim.human_num": 9, "sim.predict_method": "truth"})), 425, 0, 4, 240, "pred_h9_rand_truthobs_stride2_r0")
    trace("CrowdSimVarNum-v0", dict(NON_RAND, **dict(S2, **{"sim.human_num": 10})), 425, 0, 1, 260, "varnum_h10_stride2_test_r0")
    trace("CrowdSimPred-v0", dict(NON_RAND, **dict(S2, **{"sim.human_num": 8, "sim.predict_method": "const_vel", "humans.policy": "social_force"})), 425, 0, 1, 240,
          "pred_h8_sfhumans_constvel_stride2_test_r0")
    trace("CrowdSimPredRealGST-v0", dict(RAND, **dict(S2, **{"sim.human_num": 8, "sim.predict_method": "inferred"})), 425, 0, 1, 200, "predgst_h8_rand_stride2_test_r0")


if __name__ == "__main__":
    what = _ARGV or ["env"]
    if "env" in what:
        env_goldens()
    only = [w[5:] for w in what if w.startswith("only:")]
    if only:                # python make_golden.py only:<substring of the tag>
        _all2 = trace

        def trace(env_name, over, seed, rank, nenv, steps, tag, phase=None):  # noqa: F811
            if any(o in tag for o in only):
                _all2(env_name, over, seed, rank, nenv, steps, tag, phase)
        env_goldens()
    if "env-test" in what:  # only the test-phase traces (the train-phase fixtures stay byte-identical)
        _all = trace

        def trace(env_name, over, seed, rank, nenv, steps, tag, phase=None):  # noqa: F811
            if nenv == 1:
                _all(env_name, over, seed, rank, nenv, steps, tag, phase)
        env_goldens()
    if "policy" in what:
        import make_golden_policy
        make_golden_policy.main()
