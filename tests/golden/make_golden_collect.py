#!/usr/bin/env python3
"""Golden traces of the reference's data-collection env (CrowdSimVarNumCollect-v0, collect_data.py) -- build container only.

    python tests/golden/make_golden_collect.py        # rewrites tests/golden/collect_*.npz

The env is driven exactly like collect_data.py drives it: robot.policy = 'orca', a dummy zero action per step, one env of a
`nenv`-env vec-env (thisSeed = seed + rank, phase 'train').  Recorded: the `pred_info` observation (frame id, prediction id, absolute
position of every visible human; cast to float32 like the vec-env buffers), the info code, the robot's state (its goal is re-drawn
whenever it reaches it) and the text lines collect_data.py would write for this env.

There is no single-env trace: make_env turns ONE env into phase 'test' (rl/networks/envs.py:54-58), and the first step of this env
class in that phase raises AttributeError in the reference (crowd_sim_var_num.py:388 -> :225 reads self.human_visibility, which
crowd_sim_var_num_collect.py's generate_ob never assigns) -- tried here with env.nenv = 1, env.phase = 'test'.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402

R.install()
INFO_CODE = {"Nothing": 0, "Timeout": 1, "Collision": 2, "ReachGoal": 3, "Danger": 4}


def trace(over, seed, rank, nenv, steps, tag):
    import crowd_sim.envs as E
    cfg = R.make_config(**over)
    cfg.robot.policy = "orca"                       # collect_data.py:14
    cfg.args.env_name = "CrowdSimVarNumCollect-v0"
    env = E.CrowdSimVarNumCollect()
    env.configure(cfg)
    env.thisSeed, env.nenv, env.phase = seed + rank, nenv, "train"
    H = cfg.sim.human_num + cfg.sim.human_num_range
    rec = {k: [] for k in ("pred_info", "info", "reward", "done", "robot_state")}
    lines = []

    def grab(ob):
        p = np.asarray(ob["pred_info"], dtype=np.float32)
        assert p.shape == (H, 4)
        return p

    ob = grab(env.reset())
    reset_pred = ob.copy()
    reset_robot = np.array(env.robot.get_full_state_list(), dtype=np.float64)
    pred_interval = int(cfg.data.pred_timestep // cfg.env.time_step)
    for t in range(steps):
        if t % pred_interval == 0:                   # collect_data.py:55-61
            rows = ob[np.logical_not(np.isinf(ob[:, -1]))].reshape(-1, 4).tolist()
            lines.extend("%s\t%s\t%s\t%s" % (str(r[0]), str(r[1]), str(r[2]), str(r[3])) for r in rows)
        o, reward, done, info = env.step(np.zeros(2))
        ob = grab(o)
        rec["pred_info"].append(ob)
        rec["info"].append(INFO_CODE[type(info["info"]).__name__])
        rec["reward"].append(float(reward))
        rec["done"].append(bool(done))
        rec["robot_state"].append(np.array(env.robot.get_full_state_list(), dtype=np.float64))
    out = {k: np.array(v) for k, v in rec.items()}
    out["reset_pred_info"] = reset_pred
    out["reset_robot"] = reset_robot
    out["lines"] = np.array("\n".join(lines))
    out["meta"] = np.array(json.dumps(dict(env_name="CrowdSimVarNumCollect-v0", over=over, seed=seed, rank=rank, nenv=nenv, steps=steps)))
    path = os.path.join(HERE, "collect_%s.npz" % tag)
    np.savez_compressed(path, **out)
    ids = out["pred_info"][:, :, 1]
    print("%-24s steps=%d infos=%s max pred id=%d lines=%d goal changes=%d -> %s (%.0f KB)" % (
        tag, steps, np.bincount(out["info"], minlength=5).tolist(), int(ids.max()), len(lines),
        int(np.sum(np.any(np.diff(out["robot_state"][:, 4:6], axis=0) != 0, axis=1))), os.path.basename(path), os.path.getsize(path) / 1024))


NON_RAND = {"env.randomize_attributes": False, "humans.random_goal_changing": False, "humans.end_goal_changing": True,
            "sim.predict_method": "none", "env.use_wrapper": False}
RAND = {"env.randomize_attributes": True, "humans.random_goal_changing": True, "humans.end_goal_changing": True,
        "sim.predict_method": "none", "env.use_wrapper": False}

if __name__ == "__main__":
    trace(dict(NON_RAND, **{"sim.human_num": 20}), 425, 0, 5, 600, "h20_nonrand_r0")
    trace(dict(RAND, **{"sim.human_num": 10}), 77, 3, 5, 600, "h10_rand_r3")
