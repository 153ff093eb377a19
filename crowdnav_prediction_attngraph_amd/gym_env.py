"""Single-environment gym objects of the reference's `crowd_sim` package (crowd_sim/__init__.py:8-26 registry,
crowd_sim/envs/{crowd_sim_var_num,crowd_sim_pred,crowd_sim_pred_real_gst}.py) as E = 1 views over the device simulator.

`make_env` (rl/networks/envs.py:36-94) drives exactly this surface: `gym.make(id)`, `env.configure(config)`, the settable
attributes `thisSeed`, `nenv`, `phase`, `render_axis`, `test_case`, `env.seed(s)`, `observation_space` (a Dict with keys in
sorted order), `action_space` (Box(2)), `reset() -> dict`, `step(action) -> (dict, float, bool, {'info': obj})`, `talk2Env(data)`.
The object owns a cn_env_batch of ONE env created with auto_reset = 0 (a single gym env does not reset itself; the vec-env
layer does): the terminal observation is returned like the reference does.  It exists for drop-in completeness and for
debugging single episodes -- throughput comes from `make_vec_envs`, which keeps all envs in one batch.

gym itself is not a dependency: `registry` / `make` mirror `gym.envs.registration` for the three accelerated ids.
"""
import numpy as np
import torch

from . import _abi as A
from . import info as I
from .config import to_env_config
from .hip import HipEnvBatch
from .policy import make_spaces

_KEYS = ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num", "visible_masks")


class _SingleCrowdSim(object):
    env_id = None
    metadata = {"render.modes": ["human"]}
    reward_range = (-float("inf"), float("inf"))
    spec = None

    def __init__(self):
        self.config = None
        self.thisSeed = None       # rl/networks/envs.py:52
        self.nenv = None           # :53
        self.phase = None          # :54-57
        self.test_case = None      # :62-63
        self.render_axis = None
        self.observation_space = None
        self.action_space = None
        self.case_counter = None
        self._env = None
        self._key = None
        self.gst_out_traj = None

    @property
    def unwrapped(self):
        return self

    # crowd_sim/envs/crowd_sim.py:88-202 configure(): everything the device simulator needs is read by to_env_config at reset
    def configure(self, config):
        self.config = config
        g = lambda ns, name, default: getattr(getattr(config, ns, None), name, default)  # noqa: E731
        H = int(g("sim", "human_num", 20)) + int(g("sim", "human_num_range", 0))
        D = 2 if self.env_id == "CrowdSimVarNum-v0" else 2 * (int(g("sim", "predict_steps", 5)) + 1)
        self.human_num, self.max_human_num = int(g("sim", "human_num", 20)), H
        self.time_step, self.time_limit = g("env", "time_step", 0.25), g("env", "time_limit", 50)
        # crowd_sim_var_num.py:37-58 / crowd_sim_pred.py:38-58 / crowd_sim_pred_real_gst.py:30-62
        self.observation_space, self.action_space = make_spaces(H, D, with_masks=self.env_id != "CrowdSimPred-v0")

    def seed(self, seed=None):
        return [seed]

    def _ensure(self, phase):
        if self.config is None:
            raise AttributeError("configure(config) has to be called before reset()")  # the reference: 'robot has to be set!'
        if not torch.cuda.is_available():
            raise A.CnError("the crowd simulator runs on MI355X only (no CPU fallback)")
        seed = int(self.thisSeed if self.thisSeed is not None else 0)
        nenv = int(self.nenv if self.nenv is not None else 1)
        key = (phase, seed, nenv)
        if self._env is None or key != self._key:
            if self._env is not None:
                self._env.close()
            cfg = to_env_config(self.config, self.env_id, nenv, phase)
            cfg.auto_reset = 0
            self._env = HipEnvBatch(cfg, 1, seed)
            self._key = key
        return self._env

    def _export(self, obs):
        out = {}
        for k in self.observation_space.spaces:
            a = obs[k][0].cpu().numpy()
            out[k] = a.astype(bool) if k == "visible_masks" else a
        return out

    def reset(self, phase="train", test_case=None):
        """crowd_sim_var_num.py:303-363: `self.phase` / `self.test_case` override the arguments when set."""
        if self.phase is not None:
            phase = self.phase
        if self.test_case is not None:
            test_case = self.test_case
        assert phase in ["train", "val", "test"]
        env = self._ensure(phase)
        if test_case is not None:
            env.set_case_counters(torch.tensor([int(test_case)], dtype=torch.int64))    # :316-318
        return self._export(env.reset())

    def step(self, action, update=True):
        if self._env is None:
            raise A.CnError("step() before reset()")
        if not update:
            raise NotImplementedError("update=False (one-step lookahead without committing the state) is not on the accelerated path")
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, 2)).to(self._env.device)
        obs, reward, done, info, _, _ = self._env.step(a)
        code = int(info.cpu()[0])
        if code == 4 and self._env.cfg.phase in (1, 2):
            inf = I.from_code(code, float(self._env.get_danger_min_dist().cpu()[0]))
        else:
            inf = I.from_code(code)
        return self._export(obs), float(reward.cpu()[0]), bool(done.cpu()[0]), {"info": inf}

    def talk2Env(self, data):
        self.gst_out_traj = data          # crowd_sim_pred_real_gst.py:64-74 (render aid)
        return True

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope of the accelerated path (use the reference env to visualise)")

    def close(self):
        if self._env is not None:
            self._env.close()
            self._env = None

    def __repr__(self):
        return "<%s instance (MI355X, E=1)>" % type(self).__name__


class CrowdSimVarNum(_SingleCrowdSim):
    env_id = "CrowdSimVarNum-v0"


class CrowdSimPred(_SingleCrowdSim):
    env_id = "CrowdSimPred-v0"


class CrowdSimPredRealGST(_SingleCrowdSim):
    env_id = "CrowdSimPredRealGST-v0"


class CrowdSimVarNumCollect(_SingleCrowdSim):
    """crowd_sim/envs/crowd_sim_var_num_collect.py: the GST dataset generator.  Observation = {'pred_info': [H, 4]} (frame id, prediction
    id, absolute px, py; +inf for humans the robot does not see); the robot is ORCA-driven (collect_data.py:14 sets robot.policy = 'orca')."""
    env_id = "CrowdSimVarNumCollect-v0"

    def configure(self, config):
        super().configure(config)
        from .collect import _Box, _DictSpace
        self.observation_space = _DictSpace({"pred_info": _Box((self.max_human_num, 4))})     # crowd_sim_var_num_collect.py:36

    def _ensure(self, phase):
        return super()._ensure("train" if phase is None else phase)

    def _export(self, obs):
        return {"pred_info": obs["spatial_edges"][0].cpu().numpy()}


# crowd_sim/__init__.py:8-26 -- id -> entry point, for the ids that are on the accelerated path
registry = {cls.env_id: cls for cls in (CrowdSimVarNum, CrowdSimPred, CrowdSimPredRealGST, CrowdSimVarNumCollect)}
_NOT_ACCELERATED = ("CrowdSim-v0", "rosTurtlebot2iEnv-v0")


def make(env_id):
    """gym.make for the accelerated ids."""
    if env_id in _NOT_ACCELERATED:
        raise NotImplementedError("%s is registered by the reference but is outside the accelerated path (SURVEY.md 8f)" % env_id)
    if env_id not in registry:
        raise KeyError("unknown env id %r (accelerated ids: %s)" % (env_id, sorted(registry)))
    return registry[env_id]()
