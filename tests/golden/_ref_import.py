"""Import helper for the *Python reference* -- usable ONLY in the build container (needs /root/reference).

Nothing in the -m gpu tests, smoke() or bench.py imports this module: the reference does not exist on the
GPU box.  It is used by tests/golden/make_golden.py to (re)generate the committed .npz fixtures.

Recipe: stub `gym` / `baselines`, alias the removed `np.bool`, fix sys.argv before
`crowd_nav.configs.config` is imported (it parses argv at class-definition time, config.py:11), and provide an
`rvo2` module -- rvo2 (Python-RVO2) is third-party, absent from /root/reference and not installed, so the shim
routes PyRVOSimulator onto the oracle's fp32 RVO2 restatement (oracle/crowdsim_oracle.c).
"""
import os
import sys
import types
from collections import OrderedDict

import numpy as np

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Box:
    def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class _Dict:
    def __init__(self, d):
        self.spaces = OrderedDict(sorted(d.items()))


class _Env:
    def seed(self, s=None):
        return [s]

    def close(self):
        pass


class PyRVOSimulator:
    """The 8 methods orca.py:80-114 uses, over oracle.orca_velocity (agent 0 only: it is the only one read)."""

    def __init__(self, timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed, velocity=(0, 0)):
        self.time_step = timeStep
        self.agents = []
        self.calls = 0

    def addAgent(self, pos, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed, velocity):
        self.agents.append(dict(pos=tuple(pos), nd=neighborDist, mn=maxNeighbors, th=timeHorizon, r=radius,
                                ms=maxSpeed, vel=tuple(velocity), pref=(0.0, 0.0)))
        return len(self.agents) - 1

    def getNumAgents(self):
        return len(self.agents)

    def setAgentPosition(self, i, pos):
        self.agents[i]["pos"] = tuple(pos)

    def setAgentVelocity(self, i, vel):
        self.agents[i]["vel"] = tuple(vel)

    def setAgentPrefVelocity(self, i, vel):
        self.agents[i]["pref"] = tuple(vel)

    def doStep(self):
        from oracle import oracle as O
        a = self.agents[0]
        others = [[o["pos"][0], o["pos"][1], o["vel"][0], o["vel"][1], o["r"]] for o in self.agents[1:]]
        v = O.orca_velocity((a["pos"][0], a["pos"][1], a["vel"][0], a["vel"][1], a["r"], a["ms"], a["pref"][0], a["pref"][1]),
                            others, neighbor_dist=a["nd"], time_horizon=a["th"], time_step=self.time_step,
                            max_neighbors=a["mn"])
        a["vel"] = (float(v[0]), float(v[1]))
        self.calls += 1

    def getAgentVelocity(self, i):
        assert i == 0
        return self.agents[0]["vel"]


def install(argv=("x", "--no-cuda")):
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    if REF not in sys.path:
        sys.path.insert(1, REF)
    sys.argv = list(argv)
    if not hasattr(np, "bool"):
        np.bool = np.bool_
    spaces = _mod("gym.spaces", Box=_Box, Dict=_Dict)
    _mod("gym.spaces.box", Box=_Box)
    _mod("gym.spaces.dict", Dict=_Dict)
    reg = _mod("gym.envs.registration", register=lambda **kw: None)
    envs = _mod("gym.envs", registration=reg)
    _mod("gym", Env=_Env, Wrapper=object, ObservationWrapper=object, spaces=spaces, envs=envs)

    class _VecEnvWrapper:
        def __init__(self, venv, observation_space=None, action_space=None):
            self.venv = venv

    _mod("baselines", bench=_mod("baselines.bench", Monitor=lambda env, *a, **k: env), logger=_mod("baselines.logger"))
    _mod("baselines.common")
    _mod("baselines.common.atari_wrappers", make_atari=None, wrap_deepmind=None)
    _mod("baselines.common.vec_env", VecEnvWrapper=_VecEnvWrapper)
    _mod("baselines.common.vec_env.vec_env", VecEnv=object, CloudpickleWrapper=object, clear_mpi_env_vars=None)
    _mod("baselines.common.vec_env.util", dict_to_obs=None, obs_space_info=None, obs_to_dict=None, copy_obs_dict=None)
    _mod("baselines.common.vec_env.vec_normalize", VecNormalize=object)
    _mod("rvo2", PyRVOSimulator=PyRVOSimulator)


def make_config(**over):
    """Fresh Config class copy with overrides like {'sim.human_num': 5, 'env.randomize_attributes': False}."""
    from crowd_nav.configs.config import Config, BaseConfig
    import copy

    cfg = Config()
    # Config uses class attributes shared process-wide; deep-copy the namespaces onto the instance
    for name in dir(Config):
        v = getattr(Config, name)
        if isinstance(v, BaseConfig):
            setattr(cfg, name, copy.deepcopy(v))
    cfg.args = copy.deepcopy(Config.args)
    for k, v in over.items():
        ns, attr = k.split(".")
        setattr(getattr(cfg, ns), attr, v)
    return cfg
