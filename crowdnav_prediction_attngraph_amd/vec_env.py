"""make_vec_envs / BatchedCrowdSim -- the vec-env object of rl/networks/envs.py:97-140, with all E envs resident on
one GPU instead of one OS process per env (shmem_vec_env.py).

Contract kept: `.observation_space.spaces`, `.action_space`, `.reset() -> {key: Tensor[E,...] on device}`,
`.step(Tensor[E,2]) -> (obs, FloatTensor[E,1] on CPU, ndarray[E] bool, list[E] of dict)`, `.talk2Env`, `.render`, `.close`,
per-env seeds thisSeed = seed + global_env_index, auto-reset on done with the reset observation returned, and the
bench.Monitor `info['episode'] = {'r','l','t'}` at episode end.  `step_device()` is the zero-host-sync variant used by the
fused rollout loop.
"""
import time

import numpy as np
import torch

from . import _abi as A
from . import info as I
from .config import Config, to_env_config
from .hip import HipEnvBatch
from .policy import make_spaces

_OBS_KEYS = ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num", "visible_masks")


class BatchedCrowdSim(object):
    def __init__(self, env_name, seed, num_envs, device, config=None, phase=None, pretext_wrapper=False, predictor=None):
        if not torch.cuda.is_available():
            raise A.CnError("the batched crowd simulator runs on MI355X only (no CPU fallback)")
        self.env_name = env_name
        self.num_envs = int(num_envs)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise A.CnError("device must be a GPU: the simulator has no CPU implementation (use the reference for --no-cuda)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        config = config if config is not None else Config()
        if phase is None:
            phase = "train" if num_envs > 1 else "test"   # rl/networks/envs.py:55-58
        world, rank = 1, 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                world, rank = dist.get_world_size(), dist.get_rank()
        except Exception:
            pass
        self.cfg = to_env_config(config, env_name, self.num_envs * world, phase)
        self._env = HipEnvBatch(self.cfg, self.num_envs, int(seed if seed is not None else 0), first_env_index=rank * self.num_envs,
                                device=self.device)
        self.human_num, self.edge_width = self._env.H, self._env.D
        has_masks = env_name != "CrowdSimPred-v0"     # crowd_sim_pred.py:38-58 has no visible_masks key
        self.observation_space, self.action_space = make_spaces(self.human_num, self.edge_width, with_masks=has_masks)
        self._keys = [k for k in _OBS_KEYS if k in self.observation_space.spaces]
        self._t0 = time.time()
        self._closed = False
        # VecPretextNormalize (GST predictions written into spatial_edges[:, :, 2:], social penalty, distance sort)
        self._pretext = None
        if pretext_wrapper:
            if env_name != "CrowdSimPredRealGST-v0":
                raise ValueError("pretext_wrapper=True goes with CrowdSimPredRealGST-v0 (config.py:162-165)")
            from .gst import PretextProcessor, load_predictor
            pred = predictor if predictor is not None else load_predictor(config, self.device)
            data = getattr(config, "data", None)
            interval = int(float(getattr(data, "pred_timestep", self.cfg.time_step)) // float(self.cfg.time_step))   # vec_pretext_normalize.py:56
            self._pretext = PretextProcessor(pred.to(self.device), self.num_envs, self.human_num, int(self.cfg.predict_steps), float(self.cfg.robot_radius),
                                             float(self.cfg.human_radius), float(self.cfg.collision_penalty), self.device, pred_interval=interval)

    # ---- device-level API (no host synchronisation) ----
    def _apply_pretext(self, obs, reward):
        se, reward = self._pretext.process(obs, reward)
        obs = dict(obs)
        obs["spatial_edges"] = se
        return obs, reward

    def reset_device(self):
        obs = self._env.reset()
        if self._pretext is not None:
            self._pretext.reset_buffers()
            obs, _ = self._apply_pretext(obs, torch.zeros(self.num_envs, device=self.device))
        return obs

    def step_device(self, actions):
        """-> obs dict (internal buffers, overwritten by the next step), reward [E], done [E] u8, info [E] u8, ep_return [E] f64, ep_len [E] i32.
        The Monitor episode return sums the raw env reward (before the GST social penalty), like the reference's wrapper order."""
        obs, reward, done, info, ep_ret, ep_len = self._env.step(actions)
        if self._pretext is not None:
            obs, reward = self._apply_pretext(obs, reward)
        return obs, reward, done, info, ep_ret, ep_len

    # ---- reference-compatible API ----
    def _export_obs(self, obs):
        out = {}
        for k in self._keys:
            out[k] = obs[k].to(torch.bool) if k == "visible_masks" else obs[k].clone()
        return out

    def reset(self):
        return self._export_obs(self.reset_device())

    def step(self, actions):
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), dtype=torch.float32)
        actions = actions.to(self.device, dtype=torch.float32).reshape(self.num_envs, 2)
        obs, reward, done, info, ep_ret, ep_len = self.step_device(actions)
        out = self._export_obs(obs)
        # ONE device-to-host transfer for everything the reference returns on the host (VecPyTorch.step_wait, envs.py:216-224); the GST
        # wrapper's reward (env reward + social penalty) is a tensor of its own
        reward_np, done_h, info_h, ret_h, len_h = self._env.fetch_step_outputs()
        reward_h = torch.from_numpy(reward_np.copy()) if self._pretext is None else reward.cpu()
        done_h = done_h.copy()
        md = None
        if self.cfg.phase in (1, 2) and (info_h == 4).any():  # val / test phase: Danger carries the min distance to an intruded future position
            md = self._env.get_danger_min_dist().cpu().numpy()
        # `infos` behaves like the reference's list of E dicts but builds a dict only for the envs that finished (bench.Monitor's 'episode'
        # entry, train.py:180-182) -- O(finished envs) Python per step instead of O(E)
        infos = I.LazyInfos(info_h, np.flatnonzero(done_h), ret_h, len_h, round(time.time() - self._t0, 6), md)
        return out, reward_h.reshape(self.num_envs, 1).float(), done_h, infos

    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        return self.step(self._pending)

    def talk2Env(self, data):
        return [True] * self.num_envs       # render aid only in the reference (crowd_sim_pred_real_gst.py:64-74)

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope of the accelerated path (use the reference env to visualise)")

    # ---- checkpointing (a bit-exact --resume needs the simulator state, not only the policy: train.py:105-108 restores weights only) ----
    def state_dict(self):
        sd = {"env": self._env.state_dict(), "env_name": self.env_name, "num_envs": self.num_envs}
        if self._pretext is not None:     # VecPretextNormalize's observation history (vec_pretext_normalize.py:85-101)
            sd["pretext"] = self._pretext.state_dict()
        return sd

    def load_state_dict(self, sd):
        if sd.get("env_name") != self.env_name or sd.get("num_envs") != self.num_envs:
            raise A.CnError("checkpoint is for %s x %s envs, this vec-env is %s x %d" % (sd.get("env_name"), sd.get("num_envs"), self.env_name, self.num_envs))
        if (self._pretext is not None) != ("pretext" in sd):
            raise A.CnError("checkpoint %s the prediction wrapper's history, this vec-env %s one" % (("holds" if "pretext" in sd else "lacks"),
                                                                                                     ("has" if self._pretext is not None else "has not")))
        self._env.load_state_dict(sd["env"])
        if self._pretext is not None:
            self._pretext.load_state_dict(sd["pretext"])

    def close(self):
        if not self._closed:
            self._env.close()
            self._closed = True


def make_vec_envs(env_name, seed, num_processes, gamma, log_dir, device, allow_early_resets, num_frame_stack=None, config=None,
                  ax=None, test_case=-1, wrap_pytorch=True, pretext_wrapper=False, phase=None, predictor=None):
    """Same signature as rl/networks/envs.py:97-109 (+ optional `phase`, and `predictor` to inject a GSTPredictor instead of
    loading config.pred.model_dir)."""
    if ax is not None:
        raise NotImplementedError("rendering (ax=...) is out of scope of the accelerated path")
    if num_frame_stack is not None:
        raise NotImplementedError("frame stacking applies to image observations only")
    # test_case: rl/networks/envs.py:60-63 only forwards it to the env inside `if ax:` (rendering), so without an axis the
    # reference ignores the argument -- same here.  Replaying chosen cases: HipEnvBatch.set_case_counters /
    # evaluation.evaluate_batched, or the single-env objects of gym_env (env.test_case = k).
    if env_name == "CrowdSimVarNumCollect-v0":      # the GST dataset generator (collect_data.py): its own observation dict
        from .collect import CollectVecEnv
        return CollectVecEnv(seed, num_processes, device, config=config, wrap_pytorch=wrap_pytorch)
    return BatchedCrowdSim(env_name, seed, num_processes, device, config=config, phase=phase, pretext_wrapper=pretext_wrapper, predictor=predictor)
