"""Shared helpers for the golden-vector tests (fixtures were produced by tests/golden/make_golden.py)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENV_KIND = {"CrowdSimVarNum-v0": 0, "CrowdSimPred-v0": 1, "CrowdSimPredRealGST-v0": 2}


ORACLE_ONLY = ()   # settings the HIP simulator does not implement (none left)


def env_fixtures(device=False):
    """All env traces; device=True leaves out the ones that exercise oracle-only settings (social-force humans)."""
    paths = sorted(glob.glob(os.path.join(GOLDEN, "env_*.npz")))
    return [p for p in paths if not (device and any(t in os.path.basename(p) for t in ORACLE_ONLY))]


def load(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def sim_kwargs(meta, oracle=False):
    """Reference config overrides -> the flat keyword set both the oracle and the HIP env understand (oracle=True adds the
    settings only the oracle implements)."""
    over = meta["over"]
    extra = {}
    if over.get("humans.policy", "orca") == "social_force":
        extra["humans_policy"] = 1
    if over.get("sim.human_num_range", 0):
        extra["human_num_range"] = int(over["sim.human_num_range"])
    if over.get("sim.predict_method", "none") == "truth":
        extra["predict_truth"] = 1
    if over.get("action_space.kinematics", "holonomic") == "unicycle":
        extra["kinematics"] = 1
    if "robot.FOV" in over:
        extra["robot_fov"] = float(over["robot.FOV"])
    if "humans.FOV" in over:
        extra["human_fov"] = float(over["humans.FOV"])
    if "env.val_size" in over:
        extra["val_size"] = int(over["env.val_size"])
    if "data.pred_timestep" in over:      # crowd_sim.py:180 (env.time_step is 0.25 in every trace)
        extra["pred_interval"] = int(float(over["data.pred_timestep"]) // 0.25)
    return dict(extra,
        human_num=int(over.get("sim.human_num", 20)),
        env_kind=ENV_KIND[meta["env_name"]],
        randomize_attributes=int(bool(over.get("env.randomize_attributes", True))),
        random_goal_changing=int(bool(over.get("humans.random_goal_changing", True))),
        end_goal_changing=int(bool(over.get("humans.end_goal_changing", True))),
        sort_humans=int(bool(meta.get("sort_humans", True))),
        nenv=int(meta["nenv"]),
        phase={"train": 0, "val": 1, "test": 2}[meta["phase"]] if meta.get("phase") else (0 if meta["nenv"] > 1 else 2),
        robot_policy={"orca": 1, "social_force": 2}.get(over.get("robot.policy", "selfAttn_merge_srnn"), 0),
        robot_visible=int(bool(over.get("robot.visible", False))),
    )
