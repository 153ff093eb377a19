"""ctypes binding of oracle/libcrowdsim_oracle.so (the C restatement in crowdsim_oracle.c).

TEST INFRASTRUCTURE ONLY -- see crowdsim_oracle.h for the reference file:line map.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcrowdsim_oracle.so")

MAX_HUMANS = 64
MAX_PRED = 8
ENV_VARNUM, ENV_PRED, ENV_PRED_GST, ENV_COLLECT = 0, 1, 2, 3
PHASE_TRAIN, PHASE_VAL, PHASE_TEST = 0, 1, 2
INFO_NAMES = {0: "Nothing", 1: "Timeout", 2: "Collision", 3: "ReachGoal", 4: "Danger"}


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "crowdsim_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcrowdsim_oracle.so"])
    return _LIB_PATH


class OrcConfig(C.Structure):
    _fields_ = [
        ("human_num", C.c_int32), ("predict_steps", C.c_int32), ("env_kind", C.c_int32),
        ("randomize_attributes", C.c_int32), ("random_goal_changing", C.c_int32),
        ("end_goal_changing", C.c_int32), ("sort_humans", C.c_int32), ("phase", C.c_int32),
        ("nenv", C.c_int32), ("val_size", C.c_uint32), ("test_size", C.c_uint32), ("robot_policy", C.c_int32),
        ("robot_visible", C.c_int32), ("max_placement_attempts", C.c_int32),
        ("time_step", C.c_double), ("time_limit", C.c_double),
        ("success_reward", C.c_double), ("collision_penalty", C.c_double),
        ("discomfort_dist", C.c_double), ("discomfort_penalty_factor", C.c_double),
        ("circle_radius", C.c_double), ("arena_size", C.c_double),
        ("human_radius", C.c_double), ("human_v_pref", C.c_double),
        ("robot_radius", C.c_double), ("robot_v_pref", C.c_double), ("sensor_range", C.c_double),
        ("goal_change_chance", C.c_double), ("end_goal_change_chance", C.c_double),
        ("orca_neighbor_dist", C.c_double), ("orca_safety_space", C.c_double),
        ("orca_time_horizon", C.c_double), ("orca_time_horizon_obst", C.c_double),
        ("sf_A", C.c_double), ("sf_B", C.c_double), ("sf_KI", C.c_double),
        ("humans_policy", C.c_int32), ("human_num_range", C.c_int32), ("kinematics", C.c_int32), ("predict_truth", C.c_int32),
        ("robot_fov", C.c_double), ("human_fov", C.c_double),
        ("pred_interval", C.c_int32), ("pad0", C.c_int32),
    ]


class OrcObs(C.Structure):
    _fields_ = [
        ("robot_node", C.c_float * 7), ("temporal_edges", C.c_float * 2),
        ("spatial_edges", C.c_float * (MAX_HUMANS * 2 * (MAX_PRED + 1))),
        ("detected_human_num", C.c_float), ("visible_masks", C.c_uint8 * MAX_HUMANS),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_env_new.restype = C.c_void_p
        L.orc_env_new.argtypes = [C.POINTER(OrcConfig), C.c_int64]
        L.orc_env_free.argtypes = [C.c_void_p]
        L.orc_env_set_case_counter.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_env_reset.argtypes = [C.c_void_p, C.POINTER(OrcObs)]
        L.orc_env_step.restype = C.c_int
        L.orc_env_step.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(OrcObs), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_env_step_autoreset.restype = C.c_int
        L.orc_env_step_autoreset.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(OrcObs), C.POINTER(C.c_double),
                                             C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_config_default.argtypes = [C.POINTER(OrcConfig)]
        L.orc_env_human_count.argtypes = [C.c_void_p]
        L.orc_mt_randint.restype = C.c_int64
        L.orc_mt_randint.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.orc_mt_seed.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_mt_double.restype = C.c_double
        L.orc_mt_double.argtypes = [C.c_void_p]
        L.orc_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_exp.restype = C.c_double
        L.orc_exp.argtypes = [C.c_double]
        L.orc_log.restype = C.c_double
        L.orc_log.argtypes = [C.c_double]
        L.orc_in_fov.restype = C.c_int
        L.orc_in_fov.argtypes = [C.c_int] + [C.c_double] * 8
        L.orc_mt_normal.restype = C.c_double
        L.orc_mt_normal.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        fp = C.POINTER(C.c_float)
        L.orc_orca_velocity.restype = C.c_int
        L.orc_orca_velocity.argtypes = [C.c_float] * 9 + [C.c_int, C.c_float, C.c_float, C.c_int, fp, fp, fp, fp, fp,
                                                          fp, fp, fp, C.POINTER(C.c_int)]
        L.orc_gae.argtypes = [C.c_int, C.c_int, fp, fp, fp, C.c_double, C.c_double, fp]
        L.orc_env_batch_step.argtypes = [C.POINTER(C.c_void_p), C.c_int, fp, fp, fp, fp, fp, C.POINTER(C.c_uint8), fp,
                                         C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        assert L.orc_sizeof_config() == C.sizeof(OrcConfig), (L.orc_sizeof_config(), C.sizeof(OrcConfig))
        assert L.orc_sizeof_obs() == C.sizeof(OrcObs)
        _lib = L
    return _lib


def default_config(**over):
    cfg = OrcConfig()
    lib().orc_config_default(C.byref(cfg))
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    if cfg.human_num + cfg.human_num_range > MAX_HUMANS or cfg.human_num_range >= max(cfg.human_num, 1):
        raise ValueError("human_num + human_num_range must be <= %d and human_num > human_num_range (crowd_sim.py:158)" % MAX_HUMANS)
    if cfg.predict_truth and cfg.env_kind != ENV_PRED:
        raise NotImplementedError("predict_method='truth': CrowdSimPred-v0 only")
    if cfg.robot_visible and cfg.env_kind == ENV_PRED and not cfg.predict_truth:
        raise ValueError("robot.visible with sim.predict_method='const_vel': the reference itself fails there (crowd_sim_var_num.py:174 "
                         "assigns the H previous human states to H + 1 rows)")
    if cfg.env_kind == ENV_COLLECT and (cfg.human_num_range or cfg.kinematics or cfg.phase != 0 or cfg.robot_policy != 1):
        raise NotImplementedError("CrowdSimVarNumCollect-v0: fixed crowd size, holonomic ORCA-driven robot, phase train (what collect_data.py runs)")
    if cfg.phase == PHASE_VAL and cfg.env_kind != ENV_PRED:
        raise ValueError("phase 'val': CrowdSimPred-v0 only (the other env classes fail at crowd_sim_var_num.py:501: self.human_future_traj "
                         "is only ever assigned by calc_human_future_traj, which they call in the test phase only)")
    if cfg.kinematics == 1 and cfg.env_kind == ENV_COLLECT:
        raise NotImplementedError("unicycle robot: not with CrowdSimVarNumCollect-v0")
    if cfg.kinematics == 1 and cfg.robot_policy != 0:
        raise NotImplementedError("unicycle robot: the ORCA / social-force robot policies return ActionXY (holonomic only)")
    return cfg


def obs_width(cfg):
    if cfg.env_kind == ENV_COLLECT:
        return 4
    return 2 if cfg.env_kind == ENV_VARNUM else 2 * (cfg.predict_steps + 1)


def _obs_to_dict(o, cfg):
    H, D = cfg.human_num + cfg.human_num_range, obs_width(cfg)   # rows = the largest crowd the config allows
    return {
        "robot_node": np.array(o.robot_node, dtype=np.float32).reshape(1, 7),
        "temporal_edges": np.array(o.temporal_edges, dtype=np.float32).reshape(1, 2),
        "spatial_edges": np.ctypeslib.as_array(o.spatial_edges)[: H * D].astype(np.float32).reshape(H, D).copy(),
        "detected_human_num": np.array([o.detected_human_num], dtype=np.float32),
        "visible_masks": np.array(o.visible_masks[:H], dtype=bool),
    }


class OracleEnv:
    """One scalar env (mirrors a reference env wrapped by bench.Monitor + the vec-env auto-reset)."""

    def __init__(self, cfg, this_seed):
        self.cfg = cfg
        self._L = lib()
        self._h = C.c_void_p(self._L.orc_env_new(C.byref(cfg), int(this_seed)))
        self._obs = OrcObs()

    def __del__(self):
        try:
            self._L.orc_env_free(self._h)
        except Exception:
            pass

    def set_case_counter(self, value):
        self._L.orc_env_set_case_counter(self._h, int(value))

    @property
    def human_count(self):
        """len(env.humans) right now (varies when human_num_range > 0 or the robot is a unicycle)."""
        return self._L.orc_env_human_count(self._h)

    def reset(self):
        self._L.orc_env_reset(self._h, C.byref(self._obs))
        return _obs_to_dict(self._obs, self.cfg)

    def step(self, action, autoreset=False):
        a = (C.c_float * 2)(float(np.float32(action[0])), float(np.float32(action[1])))
        r, info = C.c_double(), C.c_int()
        if autoreset:
            epr, epl, md = C.c_double(), C.c_int(), C.c_double()
            done = self._L.orc_env_step_autoreset(self._h, a, C.byref(self._obs), C.byref(r), C.byref(info),
                                                  C.byref(epr), C.byref(epl), C.byref(md))
            extra = {"episode": {"r": round(epr.value, 6), "l": epl.value}} if done else {}
            extra["min_dist"] = md.value
        else:
            md = C.c_double()
            done = self._L.orc_env_step(self._h, a, C.byref(self._obs), C.byref(r), C.byref(info), C.byref(md))
            extra = {"min_dist": md.value}
        return _obs_to_dict(self._obs, self.cfg), r.value, bool(done), dict(info=info.value, **extra)

    # raw state access for tests (layout of OrcEnv is private; expose via small helpers instead)


def orca_velocity(self_state, others, neighbor_dist=10.0, time_horizon=5.0, time_step=0.25, max_neighbors=None,
                  want_lines=False):
    """self_state = (px,py,vx,vy,radius,max_speed,pref_vx,pref_vy); others = [n][5] (px,py,vx,vy,radius)."""
    L = lib()
    o = np.ascontiguousarray(np.asarray(others, dtype=np.float32).reshape(-1, 5))
    n = o.shape[0]
    cols = [np.ascontiguousarray(o[:, k]) for k in range(5)]
    fp = C.POINTER(C.c_float)
    ox, oy = C.c_float(), C.c_float()
    lines = np.zeros((max(n, 1), 4), dtype=np.float32)
    fail = C.c_int()
    s = [float(np.float32(x)) for x in self_state]
    nl = L.orc_orca_velocity(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], float(np.float32(neighbor_dist)),
                             n if max_neighbors is None else max_neighbors, float(np.float32(time_horizon)),
                             float(np.float32(time_step)), n, *[c.ctypes.data_as(fp) for c in cols],
                             C.byref(ox), C.byref(oy), lines.ctypes.data_as(fp), C.byref(fail))
    if want_lines:
        return (ox.value, oy.value), lines[:nl].copy(), fail.value
    return ox.value, oy.value


class MT:
    """numpy-legacy MT19937 stream (for RNG parity tests)."""

    def __init__(self, seed):
        self._buf = C.create_string_buffer(4 * 624 + 8)
        lib().orc_mt_seed(self._buf, seed)

    def random(self):
        return lib().orc_mt_double(self._buf)

    def randint(self, low, high):
        return lib().orc_mt_randint(self._buf, low, high, None)

    def normal(self, loc=0.0, scale=1.0):
        if not hasattr(self, "_hg"):
            self._hg, self._g = C.c_int32(0), C.c_double(0.0)
        return lib().orc_mt_normal(self._buf, C.byref(self._hg), C.byref(self._g), float(loc), float(scale), None)


def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def gae(rewards, values, masks, gamma, lam):
    """rewards [T,N], values/masks [T+1,N] float32 -> returns [T,N]."""
    r = np.ascontiguousarray(rewards, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    m = np.ascontiguousarray(masks, dtype=np.float32)
    T, N = r.shape
    out = np.zeros((T + 1, N), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib().orc_gae(T, N, r.ctypes.data_as(fp), v.ctypes.data_as(fp), m.ctypes.data_as(fp), gamma, lam, out.ctypes.data_as(fp))
    return out[:T]
