"""TEST INFRASTRUCTURE: a CPU vec-env with the contract of crowdnav_prediction_attngraph_amd.vec_env.BatchedCrowdSim whose
simulator is the C oracle (oracle/crowdsim_oracle.c), one scalar env per slot.  It exists so that the reference's own
train.py can be executed, unchanged, against the dropin/ module names in the build container (no GPU there) -- see
tests/test_dropin_train_surface.py.  Never imported by the product."""
import time

import numpy as np
import torch

from crowdnav_prediction_attngraph_amd import info as I
from crowdnav_prediction_attngraph_amd.config import Config, to_env_config
from crowdnav_prediction_attngraph_amd.policy import make_spaces
from oracle import oracle as O

_FIELDS = ("human_num", "predict_steps", "env_kind", "randomize_attributes", "random_goal_changing", "end_goal_changing", "sort_humans",
           "phase", "nenv", "robot_policy", "robot_visible", "time_step", "time_limit", "success_reward", "collision_penalty",
           "discomfort_dist", "discomfort_penalty_factor", "circle_radius", "arena_size", "human_radius", "human_v_pref", "robot_radius",
           "robot_v_pref", "sensor_range", "goal_change_chance", "end_goal_change_chance", "orca_neighbor_dist", "orca_safety_space",
           "orca_time_horizon", "orca_time_horizon_obst")


class OracleVecEnv(object):
    def __init__(self, env_name, seed, num_envs, device, config=None, phase=None):
        config = config if config is not None else Config()
        if phase is None:
            phase = "train" if num_envs > 1 else "test"
        cn = to_env_config(config, env_name, num_envs, phase)          # the same validation / field mapping as the product
        self.cfg = O.default_config(**{k: getattr(cn, k) for k in _FIELDS})
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.envs = [O.OracleEnv(self.cfg, int(seed) + i) for i in range(self.num_envs)]
        H, D = int(cn.human_num), O.obs_width(self.cfg)
        self.observation_space, self.action_space = make_spaces(H, D, with_masks=env_name != "CrowdSimPred-v0")
        self._keys = list(self.observation_space.spaces)
        self._t0 = time.time()

    def _stack(self, obs):
        out = {}
        for k in self._keys:
            a = np.stack([o[k] for o in obs])
            out[k] = torch.from_numpy(a).to(self.device)
        return out

    def reset(self):
        return self._stack([e.reset() for e in self.envs])

    def step(self, actions):
        a = actions.detach().cpu().numpy().reshape(self.num_envs, 2).astype(np.float32)
        obs, rew, done, infos = [], [], [], []
        for i, e in enumerate(self.envs):
            ob, r, d, inf = e.step(a[i], autoreset=True)
            obs.append(ob); rew.append(r); done.append(d)
            item = {"info": I.from_code(inf["info"], inf["min_dist"])}
            if d:
                item["episode"] = dict(inf["episode"], t=round(time.time() - self._t0, 6))
            infos.append(item)
        return self._stack(obs), torch.tensor(rew, dtype=torch.float32).unsqueeze(1), np.array(done, dtype=bool), infos

    def talk2Env(self, data):
        return [True] * self.num_envs

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        self.envs = []


class OracleCollectVecEnv(object):
    """Stand-in for crowdnav_prediction_attngraph_amd.collect.CollectVecEnv (CrowdSimVarNumCollect-v0, numpy observations)."""

    def __init__(self, seed, num_envs, device, config=None):
        config = config if config is not None else Config()
        cn = to_env_config(config, "CrowdSimVarNumCollect-v0", num_envs, "train")
        self.cfg = O.default_config(**{k: getattr(cn, k) for k in _FIELDS})
        self.num_envs = int(num_envs)
        self.envs = [O.OracleEnv(self.cfg, int(seed) + i) for i in range(self.num_envs)]

    def reset(self):
        return {"pred_info": np.stack([e.reset()["spatial_edges"] for e in self.envs])}

    def step(self, actions):
        outs = [e.step(np.zeros(2, np.float32)) for e in self.envs]
        return ({"pred_info": np.stack([o[0]["spatial_edges"] for o in outs])}, np.array([o[1] for o in outs], dtype=np.float32),
                np.array([o[2] for o in outs], dtype=bool), [{"info": I.from_code(o[3]["info"])} for o in outs])

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        self.envs = []


def make_vec_envs(env_name, seed, num_processes, gamma, log_dir, device, allow_early_resets, num_frame_stack=None, config=None,
                  ax=None, test_case=-1, wrap_pytorch=True, pretext_wrapper=False, phase=None, predictor=None):
    if pretext_wrapper:
        raise NotImplementedError("the oracle-backed test vec-env has no GST wrapper")
    if env_name == "CrowdSimVarNumCollect-v0":
        return OracleCollectVecEnv(seed, num_processes, device, config=config)
    return OracleVecEnv(env_name, seed, num_processes, device, config=config, phase=phase)
