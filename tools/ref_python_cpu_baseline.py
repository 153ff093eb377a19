#!/usr/bin/env python3
"""BUILD-CONTAINER ONLY (needs /root/reference): times the REAL reference Python on this container's host cores -- SURVEY 8d CPU baseline
(ii), BASELINE.md 4.1.

    python tools/ref_python_cpu_baseline.py [--seconds 20] [--out profiles/r04_reference_python_cpu.json]

What is timed: `env.step()` of the reference's own classes (crowd_sim/envs/crowd_sim_var_num.py:366-460, crowd_sim_pred.py:100-214), imported
through tests/golden/_ref_import.py (stub gym / baselines; the third-party `rvo2` module, absent here, is a shim over the oracle's C
restatement of RVO2) -- single process, and 8 forked worker processes each stepping its own env (the reference's SubprocVecEnv shape; its
published runs use 16).  Episodes end and reset inside the loop like in training (reset time is included, as in `fps` of train.py:235).
Also the reference `Policy.act` (rl/networks/model.py:56-74) on torch CPU with one thread (train.py:61) at the batch the reference trains
with (16 envs) -- the other half of BASELINE.json's metric "sim + policy fwd".

Two known biases of the rvo2 shim, in opposite directions (both stated in the output):
  * it is SLOWER per call than the Cython module (Python dict bookkeeping per setAgentPosition / setAgentVelocity): `shim_share` reports the
    fraction of the step spent inside the shim;
  * it does LESS arithmetic: the real `doStep()` computes the new velocity of EVERY agent of a human's private simulator (H agents, k-d tree),
    the shim only agent 0's -- the only one orca.py:114 reads.
Nothing here runs on the GPU box and nothing of the reference is copied: the script writes numbers only.
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)


def _make(env_name, H, randomized, seed, rank, nenv):
    import _ref_import as R
    R.install()
    import numpy as np  # noqa: F401
    import crowd_sim.envs as E
    over = {"sim.human_num": H, "env.randomize_attributes": bool(randomized), "humans.random_goal_changing": bool(randomized),
            "humans.end_goal_changing": True, "sim.predict_method": "const_vel" if env_name == "CrowdSimPred-v0" else "none"}
    cfg = R.make_config(**over)
    cfg.args.env_name = env_name
    cls = {"CrowdSimVarNum-v0": E.CrowdSimVarNum, "CrowdSimPred-v0": E.CrowdSimPred}[env_name]
    env = cls()
    env.configure(cfg)
    env.thisSeed, env.nenv, env.phase = seed + rank, nenv, "train"      # rl/networks/envs.py:49-58
    return env, R


def _run(env_name, H, randomized, rank, nenv, seconds):
    """Steps one env for ~`seconds` of wall time with a goal-seeking action; returns (steps, wall seconds, seconds inside the rvo2 shim)."""
    import numpy as np
    env, R = _make(env_name, H, randomized, 425, rank, nenv)
    shim_t = [0.0]
    cls = R.PyRVOSimulator
    if not getattr(cls, "_timed", False):
        for name in ("addAgent", "setAgentPosition", "setAgentVelocity", "setAgentPrefVelocity", "doStep", "getAgentVelocity", "__init__"):
            f = getattr(cls, name)

            def wrap(f):
                def g(*a, **k):
                    t = time.perf_counter()
                    try:
                        return f(*a, **k)
                    finally:
                        shim_t[0] += time.perf_counter() - t
                return g
            setattr(cls, name, wrap(f))
        cls._timed = True
    env.reset()
    steps = 0
    t0 = time.perf_counter()
    while True:
        gx, gy = env.robot.gx - env.robot.px, env.robot.gy - env.robot.py
        n = max(np.hypot(gx, gy), 1e-9)
        a = np.array([0.9 * gx / n + 0.3 * np.sin(0.37 * steps), 0.9 * gy / n + 0.3 * np.cos(0.23 * steps)], dtype=np.float32)
        _, _, done, _ = env.step(a)
        steps += 1
        if done:
            env.reset()
        if (steps & 7) == 0 and time.perf_counter() - t0 >= seconds:
            break
    return steps, time.perf_counter() - t0, shim_t[0]


def _worker(args):
    return _run(*args)


def _policy_rate(H, E, seconds):
    """reference Policy.act on torch CPU, one thread (train.py:61), batch E."""
    import _ref_import as R
    R.install()
    import numpy as np
    import torch
    torch.set_num_threads(1)
    from rl.networks.model import Policy
    cfg = R.make_config(**{"sim.human_num": H})
    args = cfg.args if hasattr(cfg, "args") else None
    from crowd_nav.configs.config import Config
    a = Config.args
    a.env_name, a.num_processes, a.num_mini_batch = "CrowdSimVarNum-v0", E, 1
    B = R._Box
    spaces = {"robot_node": B(shape=(1, 7)), "temporal_edges": B(shape=(1, 2)), "spatial_edges": B(shape=(H, 2)),
              "detected_human_num": B(shape=(1,)), "visible_masks": B(shape=(H,))}
    act_space = type("Box", (), {"shape": (2,)})()
    torch.manual_seed(425)
    net = Policy(spaces, act_space, base_kwargs=a, base="selfAttn_merge_srnn")
    obs = {"robot_node": torch.randn(E, 1, 7), "temporal_edges": torch.randn(E, 1, 2), "spatial_edges": torch.randn(E, H, 2),
           "detected_human_num": torch.full((E, 1), 6.0), "visible_masks": torch.ones(E, H, dtype=torch.bool)}
    hxs = {"human_node_rnn": torch.zeros(E, 1, 128), "human_human_edge_rnn": torch.zeros(E, H + 1, 256)}
    masks = torch.ones(E, 1)
    with torch.no_grad():
        for _ in range(3):
            net.act(obs, hxs, masks)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            net.act(obs, hxs, masks)
            n += 1
        dt = time.perf_counter() - t0
    return n * E / dt, n / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_reference_python_cpu.json"))
    a = ap.parse_args()
    cases = [("CrowdSimVarNum-v0", 5, True), ("CrowdSimVarNum-v0", 20, False), ("CrowdSimVarNum-v0", 50, True), ("CrowdSimPred-v0", 20, False)]
    res = []
    ctx = mp.get_context("fork")
    for env_name, H, rnd in cases:
        with ctx.Pool(1) as pool:          # a fresh process per case: the reference's Config is process-global state
            s, w, sh = pool.map(_worker, [(env_name, H, rnd, 0, 2, a.seconds)])[0]
        with ctx.Pool(a.workers) as pool:
            rs = pool.map(_worker, [(env_name, H, rnd, r, a.workers, a.seconds) for r in range(a.workers)])
        agg = sum(x[0] / x[1] for x in rs)
        rec = {"env": env_name, "humans": H, "randomized": rnd, "single_process_steps_per_s": round(s / w, 2), "single_process_shim_share": round(sh / w, 3),
               "workers": a.workers, "workers_aggregate_steps_per_s": round(agg, 2), "per_worker_steps_per_s": round(agg / a.workers, 2),
               "steps_timed_single": s}
        print(rec, flush=True)
        res.append(rec)
    with ctx.Pool(1) as pool:
        pol = pool.starmap(_policy_rate, [(20, 16, min(a.seconds, 10.0))])[0]
    out = {"what": "the reference's own Python env.step (crowd_sim_var_num.py:366-460 / crowd_sim_pred.py:100-214; rvo2 = shim over the oracle's C RVO2 "
                   "restatement) and Policy.act (rl/networks/model.py:56-74), timed in the BUILD container by tools/ref_python_cpu_baseline.py",
           "host": {"logical_cpus": os.cpu_count(), "machine": platform.machine(), "python": platform.python_version()},
           "seconds_per_measurement": a.seconds, "env_step": res,
           "policy_act_cpu_1thread_batch16": {"env_steps_per_s": round(pol[0], 1), "forwards_per_s": round(pol[1], 2), "humans": 20},
           "caveats": ["rvo2 shim is slower per call than the Cython module (see single_process_shim_share) but computes only agent 0 of each private "
                       "simulator, where the real doStep() computes all H agents: the two biases point in opposite directions",
                       "episodes reset inside the loop; reset time is included, as in train.py's fps",
                       "8 container cores; the reference's published runs use 16 worker processes (176-206 FPS incl. policy forward and PPO update)"]}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
