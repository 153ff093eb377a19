"""Config -- the fields of crowd_nav/configs/config.py the hot path reads, same nesting and names
(config.env.time_step, config.sim.human_num, config.humans.radius, config.orca.neighbor_dist, ...), so a reference
Config instance and this one are interchangeable as the `config=` argument of make_vec_envs.
Defaults follow crowd_nav/configs/config.py:16-120."""
import copy
import math
import types


def _ns(**kw):
    return types.SimpleNamespace(**kw)


class Config(object):
    def __init__(self, args=None, **overrides):
        self.args = args if args is not None else _ns(sort_humans=True, env_name="CrowdSimVarNum-v0", num_processes=16, seed=425)
        self.training = _ns(device="cuda:0")
        self.env = _ns(time_limit=50, time_step=0.25, val_size=100, test_size=500, randomize_attributes=True,
                       num_processes=getattr(self.args, "num_processes", 16), record=False, load_act=False, use_wrapper=False)
        self.reward = _ns(success_reward=10, collision_penalty=-20, discomfort_dist=0.25, discomfort_penalty_factor=10, gamma=0.99)
        self.sim = _ns(circle_radius=6 * math.sqrt(2), arena_size=6, human_num=20, human_num_range=0, predict_steps=5,
                       predict_method="none", render=False)
        self.humans = _ns(visible=True, policy="orca", radius=0.3, v_pref=1, sensor="coordinates", FOV=2.,
                          random_goal_changing=True, goal_change_chance=0.5, end_goal_changing=True, end_goal_change_chance=1.0,
                          random_radii=False, random_v_pref=False, random_unobservability=False, unobservable_chance=0.3,
                          random_policy_changing=False)
        self.robot = _ns(visible=False, policy="selfAttn_merge_srnn", radius=0.3, v_pref=1, sensor="coordinates", FOV=2, sensor_range=5)
        self.action_space = _ns(kinematics="holonomic")
        self.orca = _ns(neighbor_dist=10, safety_space=0.15, time_horizon=5, time_horizon_obst=5)
        self.sf = _ns(A=2., B=1, KI=1)
        self.data = _ns(tot_steps=40000, render=False, collect_train_data=False, num_processes=5,
                        data_save_dir="gst_updated/datasets/orca_20humans_no_rand", pred_timestep=0.25)   # config.py:129-136 (collect_data.py)
        self.pred = _ns(model_dir="gst_updated/results/100-gumbel_social_transformer-faster_lstm-lr_0.001-init_temp_0.5-edge_head_0-ebd_64-snl_1-snh_8-seed_1000_rand/sj")
        for k, v in overrides.items():
            ns, attr = k.split(".")
            setattr(getattr(self, ns), attr, v)

    def copy(self):
        return copy.deepcopy(self)


def non_randomized(args=None, **overrides):
    """The non-randomised preset of trained_models/GST_predictor_non_rand (BASELINE configs[1])."""
    o = {"env.randomize_attributes": False, "humans.random_goal_changing": False, "humans.end_goal_changing": True}
    o.update(overrides)
    return Config(args, **o)


def to_env_config(config, env_name, nenv_total, phase="train"):
    """Reference-style Config -> cn_env_config, rejecting settings the device simulator does not implement."""
    from . import _abi as A
    if env_name not in A.ENV_KINDS:
        raise NotImplementedError("env id %r is not on the accelerated path (supported: %s)" % (env_name, sorted(A.ENV_KINDS)))
    g = lambda ns, name, default: getattr(getattr(config, ns, None), name, default)  # noqa: E731
    unsupported = []
    hn, hr = int(g("sim", "human_num", 20)), int(g("sim", "human_num_range", 0))
    if not 0 <= hr < hn or hn + hr > 64:
        unsupported.append("sim.human_num_range outside [0, human_num) or human_num + human_num_range > 64")
    rv = bool(g("robot", "visible", False))
    if rv and hn + hr > 63:
        unsupported.append("robot.visible=True with human_num + human_num_range > 63")
    kin = g("action_space", "kinematics", "holonomic")
    if kin not in ("holonomic", "unicycle"):
        unsupported.append("action_space.kinematics=%r" % kin)
    hp = g("humans", "policy", "orca")
    if hp not in ("orca", "social_force"):
        unsupported.append("humans.policy=%r" % hp)
    hfov, rfov = float(g("humans", "FOV", 2.)), float(g("robot", "FOV", 2))
    if hfov <= 0.0 or rfov <= 0.0:
        unsupported.append("FOV <= 0")
    pm = g("sim", "predict_method", "const_vel")
    if env_name == "CrowdSimPred-v0" and pm not in ("const_vel", "truth"):
        unsupported.append("sim.predict_method=%r (CrowdSimPred-v0 runs with 'const_vel' or 'truth')" % pm)
    if env_name == "CrowdSimPred-v0" and pm == "const_vel" and rv:
        unsupported.append("CrowdSimPred-v0 with sim.predict_method='const_vel' and robot.visible=True (the reference itself fails there: "
                           "crowd_sim_var_num.py:174 assigns the H previous human states to H + 1 rows)")
    rp = g("robot", "policy", "selfAttn_merge_srnn")
    if rp not in ("selfAttn_merge_srnn", "srnn", "orca", "social_force"):
        unsupported.append("robot.policy=%r (the network policies, 'orca' and 'social_force' are implemented)" % rp)
    if kin == "unicycle" and (env_name == "CrowdSimVarNumCollect-v0" or rp in ("orca", "social_force")):
        unsupported.append("unicycle kinematics with an ORCA / social-force robot (those policies return ActionXY) or in CrowdSimVarNumCollect-v0")
    if env_name == "CrowdSimVarNumCollect-v0" and (rp != "orca" or hr != 0 or kin != "holonomic" or phase != "train"):
        unsupported.append("CrowdSimVarNumCollect-v0 outside collect_data.py's set-up (robot.policy='orca', fixed crowd size, holonomic, phase train)")
    if phase not in ("train", "val", "test") or (phase == "val" and env_name != "CrowdSimPred-v0"):
        unsupported.append("phase=%r (phase 'val' only runs in CrowdSimPred-v0: the other env classes fail at crowd_sim_var_num.py:501, "
                           "self.human_future_traj is only assigned in their test phase)" % phase)
    # prediction stride (crowd_sim.py:180): 0 makes the reference slice with step 0 (ValueError at its first 'truth' roll-out); above 16 the
    # device's roll-out buffer is not sized for it
    pred_interval = int(float(g("data", "pred_timestep", 0.25)) // float(g("env", "time_step", 0.25)))
    if not 1 <= pred_interval <= 16:
        unsupported.append("int(data.pred_timestep // env.time_step) = %d outside [1, 16]" % pred_interval)
    if unsupported:
        raise NotImplementedError("not implemented on the device path yet: " + "; ".join(unsupported))
    return A.default_env_config(
        human_num=hn, human_num_range=hr, kinematics=int(kin == "unicycle"), predict_steps=int(g("sim", "predict_steps", 5)), env_kind=A.ENV_KINDS[env_name],
        randomize_attributes=int(bool(g("env", "randomize_attributes", True))),
        random_goal_changing=int(bool(g("humans", "random_goal_changing", True))),
        end_goal_changing=int(bool(g("humans", "end_goal_changing", True))),
        sort_humans=int(bool(getattr(getattr(config, "args", None), "sort_humans", True))),
        predict_truth=int(env_name == "CrowdSimPred-v0" and pm == "truth"), pred_interval=pred_interval,
        phase={"train": 0, "val": 1, "test": 2}[phase], nenv=int(nenv_total), robot_policy={"orca": 1, "social_force": 2}.get(rp, 0), humans_policy=int(hp == "social_force"),
        sf_A=float(g("sf", "A", 2.)), sf_B=float(g("sf", "B", 1.)), sf_KI=float(g("sf", "KI", 1.)), robot_fov=rfov, human_fov=hfov, robot_visible=int(rv), val_size=int(g("env", "val_size", 100)), test_size=int(g("env", "test_size", 500)),
        time_step=float(g("env", "time_step", 0.25)), time_limit=float(g("env", "time_limit", 50)),
        success_reward=float(g("reward", "success_reward", 10)), collision_penalty=float(g("reward", "collision_penalty", -20)),
        discomfort_dist=float(g("reward", "discomfort_dist", 0.25)),
        discomfort_penalty_factor=float(g("reward", "discomfort_penalty_factor", 10)),
        circle_radius=float(g("sim", "circle_radius", 6 * math.sqrt(2))), arena_size=float(g("sim", "arena_size", 6)),
        human_radius=float(g("humans", "radius", 0.3)), human_v_pref=float(g("humans", "v_pref", 1)),
        robot_radius=float(g("robot", "radius", 0.3)), robot_v_pref=float(g("robot", "v_pref", 1)),
        sensor_range=float(g("robot", "sensor_range", 5)),
        goal_change_chance=float(g("humans", "goal_change_chance", 0.5)),
        end_goal_change_chance=float(g("humans", "end_goal_change_chance", 1.0)),
        orca_neighbor_dist=float(g("orca", "neighbor_dist", 10)), orca_safety_space=float(g("orca", "safety_space", 0.15)),
        orca_time_horizon=float(g("orca", "time_horizon", 5)), orca_time_horizon_obst=float(g("orca", "time_horizon_obst", 5)),
        # not a reference switch: the bound of the reference's unbounded placement loops (cn_env_config.max_placement_attempts; 0 = 65536)
        max_placement_attempts=int(g("sim", "max_placement_attempts", 0)))
