"""Dataset generation for the GST trajectory predictor: the batched counterpart of the reference's collect_data.py +
crowd_sim/envs/crowd_sim_var_num_collect.py (gym id CrowdSimVarNumCollect-v0).

The reference steps `config.data.num_processes` (5) OS-process envs with an ORCA-driven robot and a dummy action, appends the
visible humans' (frame id, prediction id, px, py) rows of every `pred_interval`-th observation to a per-env list and writes one
tab-separated text file per env.  Here the E envs are one device batch (cn_env_batch with env_kind CN_ENV_COLLECT): the `pred_info`
observations of a block of steps stay on the GPU and cross PCIe once per block; the files are byte-for-byte what the reference
writes for the same seeds (tests/test_gpu_collect.py replays the reference's own traces).
"""
import os

import numpy as np
import torch

from . import _abi as A
from . import info as I
from .config import Config, to_env_config
from .hip import HipEnvBatch

ENV_NAME = "CrowdSimVarNumCollect-v0"


class _Box(object):
    def __init__(self, shape, dtype=np.float32):
        self.shape, self.dtype = tuple(shape), dtype
        self.low, self.high = -np.inf, np.inf


class _DictSpace(object):
    def __init__(self, spaces):
        self.spaces = dict(spaces)


class CollectVecEnv(object):
    """The vec-env object collect_data.py gets from make_vec_envs(..., wrap_pytorch=False): observations are
    {'pred_info': ndarray [E, H, 4] float32} (crowd_sim_var_num_collect.py:36), the action is ignored (the robot is ORCA-driven).

    Episodes are generated in phase 'train' (seed offset 2000).  The reference's make_env switches ONE env to phase 'test'
    (rl/networks/envs.py:54-57), and in that phase the reference's own step() raises AttributeError for this env class
    (crowd_sim_var_num.py:388 -> :225 reads self.human_visibility, which crowd_sim_var_num_collect.py never assigns): there is no
    single-env behaviour to reproduce, so num_envs = 1 is refused (collect_data.py itself runs config.data.num_processes = 5 envs)."""

    def __init__(self, seed, num_envs, device, config=None, wrap_pytorch=False):
        if not torch.cuda.is_available():
            raise A.CnError("the batched crowd simulator runs on MI355X only (no CPU fallback)")
        self.num_envs = int(num_envs)
        if self.num_envs == 1:
            raise NotImplementedError("CrowdSimVarNumCollect-v0 with ONE env runs in phase 'test' in the reference (rl/networks/envs.py:54-57), where its "
                                      "step() fails (crowd_sim_var_num.py:225: no attribute 'human_visibility'); use num_envs >= 2 as collect_data.py does")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        config = config if config is not None else Config()
        self.cfg = to_env_config(config, ENV_NAME, self.num_envs, "train")
        self._env = HipEnvBatch(self.cfg, self.num_envs, int(seed), device=self.device)
        self.human_num = self._env.H
        self.observation_space = _DictSpace({"pred_info": _Box((self.human_num, 4))})
        self.action_space = _Box((2,))
        self._torch = bool(wrap_pytorch)
        self._closed = False

    # ---- device-level API: pred_info as a device tensor view of the env's observation buffer (overwritten by the next step) ----
    def reset_device(self):
        return self._env.reset()["spatial_edges"]

    def step_device(self, actions=None):
        if actions is None:
            actions = torch.zeros(self.num_envs, 2, device=self.device)
        obs, reward, done, info, _, _ = self._env.step(actions)
        return obs["spatial_edges"], reward, done, info

    # ---- reference-compatible API ----
    def _export(self, pred):
        return {"pred_info": pred.clone() if self._torch else pred.cpu().numpy()}

    def reset(self):
        return self._export(self.reset_device())

    def step(self, actions=None):
        if actions is not None and not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), dtype=torch.float32)
        if actions is not None:
            actions = actions.to(self.device, dtype=torch.float32).reshape(self.num_envs, 2)
        pred, reward, done, info = self.step_device(actions)
        infos = [{"info": I.from_code(int(c))} for c in info.cpu().numpy()]
        if self._torch:
            return self._export(pred), reward.cpu().unsqueeze(1), done.cpu().numpy().astype(bool), infos
        return self._export(pred), reward.cpu().numpy(), done.cpu().numpy().astype(bool), infos

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope of the accelerated path (use the reference env to visualise)")

    def close(self):
        if not self._closed:
            self._env.close()
            self._closed = True


def format_rows(pred_info):
    """The lines collect_data.py:75-79 writes for one observation of one env: the rows whose last column is finite, each as
    str(frame) TAB str(id) TAB str(px) TAB str(py) of the float32 values widened to Python floats."""
    p = np.asarray(pred_info, dtype=np.float32)
    rows = p[np.logical_not(np.isinf(p[:, -1]))].reshape(-1, 4).tolist()
    return ["%s\t%s\t%s\t%s" % (str(r[0]), str(r[1]), str(r[2]), str(r[3])) for r in rows]


def collect_lines(envs, tot_steps, pred_interval=1, block=256):
    """Run `tot_steps` simulation steps and return, per env, the list of text lines of every `pred_interval`-th observation
    (the observation BEFORE the step, starting with the reset observation: collect_data.py:52-66)."""
    E, H = envs.num_envs, envs.human_num
    lines = [[] for _ in range(E)]
    buf = torch.empty(block, E, H, 4, device=envs.device)
    pred = envs.reset_device()
    filled = 0

    def flush(n):
        host = buf[:n].cpu().numpy()
        for k in range(n):
            for i in range(E):
                lines[i].extend(format_rows(host[k, i]))

    for step in range(int(tot_steps)):
        if step % pred_interval == 0:
            buf[filled].copy_(pred)
            filled += 1
            if filled == block:
                flush(filled)
                filled = 0
        pred = envs.step_device()[0]
    flush(filled)
    return lines


def collectData(device, train_data, config, num_envs=None, seed=None):
    """collect_data.py:12-82 on the device batch.  Returns the directory written.  `num_envs` defaults to config.data.num_processes
    (the reference's 5); a few thousand envs fill a GPU and produce that many files per run."""
    config.robot.policy = "orca"                               # collect_data.py:14
    if getattr(config.data, "render", False):
        raise NotImplementedError("rendering is out of scope of the accelerated path")
    env_num = int(num_envs if num_envs is not None else config.data.num_processes)
    if seed is None:
        seed = np.random.randint(0, np.iinfo(np.uint32).max)   # collect_data.py:35
    envs = CollectVecEnv(seed, env_num, device, config=config)
    pred_interval = int(config.data.pred_timestep // config.env.time_step)
    tot_steps = int(config.data.tot_steps * pred_interval)
    lines = collect_lines(envs, tot_steps, pred_interval)
    envs.close()
    path = os.path.join(config.data.data_save_dir, "train" if train_data else "test")
    os.makedirs(path, exist_ok=True)
    for i in range(env_num):
        with open(os.path.join(path, str(i) + ".txt"), "w") as f:
            for item in lines[i]:
                f.write("%s\n" % item)
    return path
