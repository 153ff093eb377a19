"""-m gpu: the HIP simulator against the CPU oracle (bit-exact) and the reference golden traces, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from tests import golden_util as G  # noqa: E402


def _cfgs(**kw):
    from crowdnav_prediction_attngraph_amd import _abi as A
    from oracle import oracle as O
    return A.default_env_config(**kw), O.default_config(**kw)


def _actions(obs_host, t, E):
    """Scripted float32 actions from the (host copy of the) observation: goal seeking / wobble / drift / centre."""
    rn = obs_host["robot_node"].reshape(E, 7).astype(np.float64)
    g = rn[:, 3:5] - rn[:, 0:2]
    n = np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-9)
    mode = (np.arange(E) + t // 60) % 4
    a = np.zeros((E, 2))
    a[mode == 0] = (1.7 * g / n)[mode == 0]
    a[mode == 1] = (0.9 * g / n + 0.5 * np.array([np.sin(0.37 * t), np.cos(0.23 * t)]))[mode == 1]
    a[mode == 2] = np.array([0.12 * np.sin(0.11 * t), 0.12 * np.cos(0.07 * t)])
    a[mode == 3] = (-0.8 * rn[:, 0:2] / 6.0 + 0.5 * g / n)[mode == 3]
    return a.astype(np.float32)


def _unicycle_actions(t, E):
    """(change of speed, change of heading): accelerate / brake into reverse / exactly straight (|r| < 1e-4) / saturated."""
    mode = (np.arange(E) + t // 25) % 5
    a = np.zeros((E, 2))
    a[mode == 0] = [0.06, 0.03 * np.sin(0.3 * t)]
    a[mode == 1] = [-0.09, 0.05 * np.cos(0.17 * t)]
    a[mode == 2] = [0.02, 0.0]
    a[mode == 3] = [0.5 * np.sin(0.9 * t), -0.4]
    a[mode == 4] = [0.04 * np.cos(0.05 * t), 0.00005]
    return a.astype(np.float32)


CASES = {
    "varnum_h20_nonrand": dict(human_num=20),
    "varnum_h5_rand": dict(human_num=5, randomize_attributes=1, random_goal_changing=1),
    "varnum_h50_rand": dict(human_num=50, randomize_attributes=1, random_goal_changing=1),
    "varnum_h64_rand": dict(human_num=64, randomize_attributes=1, random_goal_changing=1, circle_radius=16.0),  # 64 humans do not fit the default circle
    "varnum_h1": dict(human_num=1),
    "varnum_unsorted": dict(human_num=10, sort_humans=0, randomize_attributes=1),
    "pred_h20_constvel": dict(human_num=20, env_kind=1),
    "pred_h10_rand": dict(human_num=10, env_kind=1, randomize_attributes=1, random_goal_changing=1),
    "predgst_h20": dict(human_num=20, env_kind=2),
    # test phase: seeds 1000 + case, 'truth' roll-out every step, Danger from the true future positions
    "varnum_h20_test": dict(human_num=20, phase=2),
    "varnum_h10_rand_test": dict(human_num=10, phase=2, randomize_attributes=1, random_goal_changing=1),
    "pred_h20_test": dict(human_num=20, env_kind=1, phase=2),
    # robot.policy = 'orca': the robot is driven by ORCA on its beliefs (the action argument is ignored)
    "varnum_h20_orcarobot": dict(human_num=20, robot_policy=1),
    "varnum_h10_rand_orcarobot_test": dict(human_num=10, robot_policy=1, phase=2, randomize_attributes=1, random_goal_changing=1),
    # robot.visible: the humans' ORCA sees the robot as one more neighbour
    "varnum_h20_robotvisible": dict(human_num=20, robot_visible=1),
    # sim.predict_method = 'truth' as the observation predictor (the paper's "oracle prediction" variant): the step is split around the
    # roll-out kernels; train and test phase, fixed and randomised humans
    "pred_h20_truthobs": dict(human_num=20, env_kind=1, predict_truth=1),
    "pred_h10_rand_truthobs": dict(human_num=10, env_kind=1, predict_truth=1, randomize_attributes=1, random_goal_changing=1),
    "pred_h10_rand_truthobs_test": dict(human_num=10, env_kind=1, predict_truth=1, phase=2, randomize_attributes=1, random_goal_changing=1),
    # a tight bound on the (in the reference unbounded) placement rejection loops: the cap path itself is bit-exact too
    "varnum_h50_rand_cap64": dict(human_num=50, randomize_attributes=1, random_goal_changing=1, max_placement_attempts=64),
    # sim.human_num_range > 0: the crowd size is drawn at reset and changes every 5 s; human_num + range observation rows
    "varnum_h15_range5": dict(human_num=15, human_num_range=5),
    "varnum_h10_rand_range4": dict(human_num=10, human_num_range=4, randomize_attributes=1, random_goal_changing=1),
    "pred_h12_range3": dict(human_num=12, human_num_range=3, env_kind=1),
    "predgst_h12_rand_range4": dict(human_num=12, human_num_range=4, env_kind=2, randomize_attributes=1, random_goal_changing=1),
    "varnum_h15_range5_test": dict(human_num=15, human_num_range=5, phase=2),
    "pred_h8_rand_range2_truthobs": dict(human_num=8, human_num_range=2, env_kind=1, predict_truth=1, randomize_attributes=1, random_goal_changing=1),
    "varnum_h12_range3_orcarobot": dict(human_num=12, human_num_range=3, robot_policy=1),
    "varnum_h12_rand_range3_robotvisible": dict(human_num=12, human_num_range=3, robot_visible=1, randomize_attributes=1),
    "varnum_h40_range24": dict(human_num=40, human_num_range=24, circle_radius=16.0),
    # action_space.kinematics = 'unicycle' (the sim2real preset): differential drive, speed integrated from the action, spin / reverse
    # penalties, 1..H humans per episode, humans at their goal get a new goal instead of being respawned
    "varnum_h5_unicycle": dict(human_num=5, kinematics=1),
    "varnum_h3_range2_unicycle": dict(human_num=3, human_num_range=2, kinematics=1),
    "varnum_h6_rand_range5_unicycle": dict(human_num=6, human_num_range=5, kinematics=1, randomize_attributes=1, random_goal_changing=1),
    "varnum_h6_rand_unicycle_test": dict(human_num=6, kinematics=1, phase=2, randomize_attributes=1, random_goal_changing=1),
    # humans.policy = 'social_force' (float64 forces, no solver) and robot.policy = 'social_force'
    "varnum_h20_sfhumans": dict(human_num=20, humans_policy=1),
    "varnum_h10_rand_sfhumans_robotvisible": dict(human_num=10, humans_policy=1, robot_visible=1, randomize_attributes=1, random_goal_changing=1),
    "pred_h12_rand_range3_sfhumans": dict(human_num=12, human_num_range=3, env_kind=1, humans_policy=1, randomize_attributes=1, random_goal_changing=1),
    "varnum_h20_sfrobot": dict(human_num=20, robot_policy=2),
    "varnum_h10_rand_sfrobot_test": dict(human_num=10, robot_policy=2, phase=2, randomize_attributes=1, random_goal_changing=1),
    "varnum_h10_sfrobot_sfhumans": dict(human_num=10, robot_policy=2, humans_policy=1),
    "varnum_h63_rand_robotvisible": dict(human_num=63, robot_visible=1, randomize_attributes=1, random_goal_changing=1, circle_radius=16.0),
    # robot.visible with 'truth' roll-outs (test phase / observation predictor): the real step hands every human its H - 1 fellows plus
    # the robot, the roll-outs only the fellows -> each private simulator is rebuilt twice per step (orca.py:80-82)
    "varnum_h10_robotvisible_test": dict(human_num=10, robot_visible=1, phase=2),
    "varnum_h8_rand_robotvisible_test": dict(human_num=8, robot_visible=1, phase=2, randomize_attributes=1, random_goal_changing=1),
    "varnum_h40_rand_robotvisible_test": dict(human_num=40, robot_visible=1, phase=2, randomize_attributes=1, random_goal_changing=1, circle_radius=16.0),
    "pred_h10_truthobs_robotvisible": dict(human_num=10, env_kind=1, predict_truth=1, robot_visible=1),
    "pred_h8_rand_truthobs_robotvisible_test": dict(human_num=8, env_kind=1, predict_truth=1, robot_visible=1, phase=2, randomize_attributes=1, random_goal_changing=1),
    "predgst_h8_rand_robotvisible_test": dict(human_num=8, env_kind=2, robot_visible=1, phase=2, randomize_attributes=1, random_goal_changing=1),
    # social-force humans with 'truth' roll-outs: the roll-outs roll SOCIAL_FORCE.predict (sf_truth_kernel)
    "varnum_h10_sfhumans_test": dict(human_num=10, humans_policy=1, phase=2),
    "varnum_h8_rand_sfhumans_robotvisible_test": dict(human_num=8, humans_policy=1, robot_visible=1, phase=2, randomize_attributes=1, random_goal_changing=1),
    "pred_h10_sfhumans_truthobs": dict(human_num=10, env_kind=1, predict_truth=1, humans_policy=1),
    "pred_h9_rand_range3_sfhumans_truthobs_test": dict(human_num=9, human_num_range=3, env_kind=1, predict_truth=1, humans_policy=1, phase=2,
                                                       randomize_attributes=1, random_goal_changing=1),
    # unicycle robot in CrowdSimPred-v0 / CrowdSimPredRealGST-v0: the Turtlebot wheel model with np.random.normal dead-band noise (legacy
    # polar Gaussian on the env's MT19937 stream), low-pass filtered in the test phase; humans at their goal are respawned
    "pred_h5_unicycle": dict(human_num=5, env_kind=1, kinematics=1),
    "pred_h6_rand_range5_unicycle_test": dict(human_num=6, human_num_range=5, env_kind=1, kinematics=1, phase=2, randomize_attributes=1, random_goal_changing=1),
    "predgst_h4_range3_unicycle": dict(human_num=4, human_num_range=3, env_kind=2, kinematics=1),
    # phase 'val' (CrowdSimPred-v0): seeds 0 + case modulo val_size, Danger from the predictions of the previous observation
    "pred_h12_constvel_val": dict(human_num=12, env_kind=1, phase=1),
    "pred_h8_rand_range2_truthobs_val": dict(human_num=8, human_num_range=2, env_kind=1, predict_truth=1, phase=1, val_size=5, randomize_attributes=1,
                                             random_goal_changing=1),
    # robot.FOV / humans.FOV < 2 (x pi): cone around the heading; what a human does not see is the dummy at (7, 7), with the config radius
    # frozen into its private simulator if that is what it was built from
    "varnum_h10_robotfov": dict(human_num=10, robot_fov=1.0),
    "varnum_h8_rand_humanfov_robotvisible": dict(human_num=8, human_fov=1.2, robot_visible=1, randomize_attributes=1, random_goal_changing=1),
    "pred_h10_fov_test": dict(human_num=10, env_kind=1, robot_fov=0.8, human_fov=1.5, phase=2),
    "varnum_h6_rand_sfhumans_fov": dict(human_num=6, humans_policy=1, human_fov=1.0, robot_fov=1.5, randomize_attributes=1, random_goal_changing=1),
    "varnum_h3_range2_unicycle_fov": dict(human_num=3, human_num_range=2, kinematics=1, robot_fov=1.0, human_fov=1.0),
    "varnum_h8_rand_fov_robotvisible_test": dict(human_num=8, robot_fov=1.0, human_fov=1.0, robot_visible=1, phase=2, randomize_attributes=1, random_goal_changing=1),
    "pred_h6_rand_unicycle_truthobs": dict(human_num=6, env_kind=1, kinematics=1, predict_truth=1, randomize_attributes=1, random_goal_changing=1),
    # round 4: data.pred_timestep = pred_interval x env.time_step (crowd_sim.py:180-181): const_vel offsets scale (crowd_sim_var_num.py:212), 'truth'
    # rolls predict_steps * pred_interval times and keeps every pred_interval-th state (:181, :206)
    "pred_h12_constvel_stride2": dict(human_num=12, env_kind=1, pred_interval=2),
    "pred_h9_rand_truthobs_stride2": dict(human_num=9, env_kind=1, predict_truth=1, pred_interval=2, randomize_attributes=1, random_goal_changing=1),
    "varnum_h10_stride3_test": dict(human_num=10, phase=2, pred_interval=3),
    "pred_h8_sfhumans_truthobs_stride2_test": dict(human_num=8, env_kind=1, predict_truth=1, humans_policy=1, pred_interval=2, phase=2),
    "predgst_h8_rand_robotvisible_stride2_test": dict(human_num=8, env_kind=2, robot_visible=1, pred_interval=2, phase=2, randomize_attributes=1, random_goal_changing=1),
    "pred_h10_constvel_stride4_val": dict(human_num=10, env_kind=1, phase=1, pred_interval=4),
    # round 6, dense crowds without a lane kernel (> 32 agents per env): the post-observation updates run as env_post_kernel on the side stream,
    # placement loops over a workgroup -- with a varying crowd size, in the test phase (truth roll-outs behind the ORCA pass), with the robot
    # visible to the humans' ORCA, with the predicted-trajectory observation; social-force humans keep the loops inside the step kernel
    "varnum_h36_rand_range8": dict(human_num=36, human_num_range=8, randomize_attributes=1, random_goal_changing=1),
    "varnum_h40_rand_test": dict(human_num=40, phase=2, randomize_attributes=1, random_goal_changing=1),
    "varnum_h40_rand_robotvisible": dict(human_num=40, robot_visible=1, randomize_attributes=1, random_goal_changing=1),
    "pred_h40_rand": dict(human_num=40, env_kind=1, randomize_attributes=1, random_goal_changing=1),
    "varnum_h40_rand_sfhumans": dict(human_num=40, humans_policy=1, randomize_attributes=1, random_goal_changing=1),
    # a jammed circle (50 humans on a circle of radius 4.5: no room for a new goal anywhere): most goal changes run to the bound, through the
    # one-wavefront passes AND several cooperative rounds -- the bound inside the first round, at 3.1 rounds, and far behind
    "varnum_h50_rand_jam_cap700": dict(human_num=50, randomize_attributes=1, random_goal_changing=1, circle_radius=4.5, max_placement_attempts=700),
    "varnum_h50_rand_jam_cap2500": dict(human_num=50, randomize_attributes=1, random_goal_changing=1, circle_radius=4.5, max_placement_attempts=2500),
    "varnum_h50_rand_jam_cap9000": dict(human_num=50, randomize_attributes=1, random_goal_changing=1, circle_radius=4.5, max_placement_attempts=9000),
}


@pytest.mark.parametrize("name", list(CASES))
def test_hip_env_matches_oracle_bit_exact(name):
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
    from oracle import oracle as O
    kw = dict(CASES[name])
    E, T, seed = 48, 260 if kw["human_num"] + kw.get("human_num_range", 0) < 36 else 110, 425
    kw["nenv"] = E
    ccfg, ocfg = _cfgs(**kw)
    env = HipEnvBatch(ccfg, E, seed)
    oenvs = [O.OracleEnv(ocfg, seed + i) for i in range(E)]
    obs = env.reset()
    keys = ["robot_node", "temporal_edges", "spatial_edges", "detected_human_num", "visible_masks"]
    host = {k: obs[k].cpu().numpy() for k in keys}
    for i, oe in enumerate(oenvs):
        ob = oe.reset()
        for k in keys:
            np.testing.assert_array_equal(host[k][i].astype(ob[k].dtype).reshape(ob[k].shape), ob[k], err_msg="reset %s env %d" % (k, i))
    n_done = 0
    infos_seen = set()
    counts_seen = set()
    for t in range(T):
        act = _unicycle_actions(t, E) if kw.get("kinematics", 0) else _actions(host, t, E)
        nd = torch.full((E, 1), -1.0, device=env.device)
        obs, rew, done, info, epr, epl = env.step(torch.from_numpy(act).to(env.device), not_done=nd)
        assert torch.equal(nd.view(-1), (done == 0).to(torch.float32)), "not_done mask t=%d" % t
        host = {k: obs[k].cpu().numpy() for k in keys}
        rew_h, done_h, info_h, epr_h, epl_h = rew.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy(), epr.cpu().numpy(), epl.cpu().numpy()
        md_h = env.get_danger_min_dist().cpu().numpy()
        cnt_h = env.get_human_counts().cpu().numpy()
        for i, oe in enumerate(oenvs):
            ob, r, d, inf = oe.step(act[i], autoreset=True)
            assert int(cnt_h[i]) == oe.human_count, "len(humans) t=%d env=%d" % (t, i)
            counts_seen.add(int(cnt_h[i]))
            assert md_h[i] == inf["min_dist"], "min_dist t=%d env=%d" % (t, i)
            assert bool(done_h[i]) == d, "done t=%d env=%d" % (t, i)
            assert int(info_h[i]) == inf["info"], "info t=%d env=%d" % (t, i)
            assert rew_h[i] == np.float32(r), "reward t=%d env=%d: %r vs %r" % (t, i, rew_h[i], r)
            if d:
                n_done += 1
                assert int(epl_h[i]) == inf["episode"]["l"]
                assert round(float(epr_h[i]), 6) == inf["episode"]["r"]
            infos_seen.add(inf["info"])
            for k in keys:
                np.testing.assert_array_equal(host[k][i].astype(ob[k].dtype).reshape(ob[k].shape), ob[k],
                                              err_msg="%s t=%d env=%d" % (k, t, i))
    assert n_done > 0
    if kw.get("human_num_range", 0) or kw.get("kinematics", 0):
        assert len(counts_seen) > 2   # the crowd really grew and shrank
    else:
        assert counts_seen == {kw["human_num"]}
    env.close()


@pytest.mark.parametrize("after", [0, 5, 100])
def test_cooperative_placement_is_bit_exact_wherever_the_helpers_join(after):
    """BASELINE configs[4]'s dense crowds run their long placement loops on four wavefronts per env (env_sim.hip, place_by_rejection<4>):
    the accepted candidate and the stream position behind it must not depend on WHEN the three helpers join.  The library reads the
    switches once per process, so the oracle comparisons above are re-run in a child with the cooperative kernel forced on for every
    crowd and the joining point at the first candidate (0), inside the first pass (5) and behind it (100)."""
    import os, subprocess, sys
    env = dict(os.environ, CN_ENV_COOP="1", CN_COOP_AFTER=str(after))
    sel = "test_hip_env_matches_oracle_bit_exact and (varnum_h20_nonrand or varnum_h5_rand or varnum_h50_rand or varnum_h64_rand or varnum_h15_range5 " \
          "or varnum_h50_rand_cap64 or pred_h10_rand or varnum_h10_rand_range4)"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    import re
    assert int(re.search(r"(\d+) passed", r.stdout).group(1)) >= 8 and "failed" not in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("path", G.env_fixtures(device=True), ids=lambda p: p.split("env_")[-1][:-4])
def test_hip_env_replays_reference_golden(path):
    """The reference's own traces (tests/golden) replayed on the GPU: flags exact, float32 obs <= 1e-6."""
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
    z, meta = G.load(path)
    kw = G.sim_kwargs(meta)
    cfg = A.default_env_config(**kw)
    env = HipEnvBatch(cfg, 1, meta["seed"], first_env_index=meta["rank"])
    obs = env.reset()
    has_masks = meta["env_name"] != "CrowdSimPred-v0"
    for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num"):
        np.testing.assert_allclose(obs[k].cpu().numpy()[0], z["reset_" + k].reshape(obs[k].shape[1:]), rtol=0, atol=1e-6)
    T = len(z["done"])
    for t in range(T):
        a = torch.from_numpy(z["actions"][t:t + 1].copy()).to(env.device)
        obs, rew, done, info, epr, epl = env.step(a)
        assert bool(done.item()) == bool(z["done"][t]), "done @%d" % t
        assert int(info.item()) == int(z["info"][t]), "info @%d" % t
        assert abs(float(rew.item()) - float(z["reward"][t])) <= 1e-6, "reward @%d" % t
        if "min_dist" in z.files:   # test-phase traces: Danger(min_dist) from the 'truth' roll-out
            assert abs(float(env.get_danger_min_dist().item()) - float(z["min_dist"][t])) <= 1e-9, "min_dist @%d" % t
        if z["done"][t]:
            assert int(epl.item()) == int(z["ep_len"][t])
            assert abs(float(epr.item()) - float(z["ep_return"][t])) <= 2e-6
        for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num"):
            np.testing.assert_allclose(obs[k].cpu().numpy()[0], z[k][t].reshape(obs[k].shape[1:]), rtol=0, atol=1e-6, err_msg="%s @%d" % (k, t))
        if has_masks:
            np.testing.assert_array_equal(obs["visible_masks"].cpu().numpy()[0].astype(bool), z["visible_masks"][t])
    env.close()


def test_episodes_do_not_depend_on_the_pregeneration_budget():
    """Episodes are generated ahead of time on a side stream, a bounded amount of work per launch, resumed in the next step.  Three budgets
    -- one human per launch, a few, everything at once -- give the same trajectories bit for bit (short episodes: the scripted robot runs
    into the crowd, so envs also reset while their next episode is only half staged and fall back to generating it in place)."""
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
    E = 96
    runs = []
    for ticks in (0, 700, 10 ** 9):
        env = HipEnvBatch(A.default_env_config(human_num=30, nenv=E, randomize_attributes=1, random_goal_changing=1, time_limit=6.0), E, 77)
        env.set_pregen_budget(ticks)
        obs = env.reset()
        tr, n_done = [], 0
        for t in range(160):
            rn = obs["robot_node"].view(E, 7)
            g = rn[:, 3:5] - rn[:, 0:2]
            a = (g / g.norm(dim=1, keepdim=True).clamp_min(1e-6)).contiguous()
            obs, rew, done, info, _, _ = env.step(a)
            n_done += int(done.sum())
            tr.append(torch.cat([obs["robot_node"].view(E, -1), obs["spatial_edges"].view(E, -1), rew.view(E, 1), done.view(E, 1).float()], 1).clone())
        runs.append(torch.stack(tr))
        assert n_done >= 4 * E
        env.close()
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])


def test_sharding_is_invariant_to_first_env_index():
    """Env i of a shard starting at k equals env k+i of one big batch (seeds keyed by the global env index)."""
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
    cfg = A.default_env_config(human_num=20, nenv=32)
    full = HipEnvBatch(cfg, 32, 425)
    part = HipEnvBatch(cfg, 8, 425, first_env_index=16)
    of, op = full.reset(), part.reset()
    torch.manual_seed(0)
    for t in range(80):
        a = torch.randn(32, 2, device=full.device)
        of = full.step(a)[0]
        op = part.step(a[16:24].contiguous())[0]
        for k in of:
            assert torch.equal(of[k][16:24], op[k]), (k, t)


def test_orca_solve_matches_oracle_bit_exact():
    from crowdnav_prediction_attngraph_amd.hip import orca_solve
    from oracle import oracle as O
    rs = np.random.RandomState(3)
    for n_other, spread in ((19, 6.0), (19, 1.2), (49, 2.0), (63, 1.5), (3, 0.5), (1, 0.3)):
        B = 96
        self_s = np.zeros((B, 8), np.float32)
        self_s[:, 0:2] = rs.uniform(-spread, spread, (B, 2))
        self_s[:, 2:4] = rs.uniform(-1, 1, (B, 2))
        self_s[:, 4] = 0.46
        self_s[:, 5] = rs.uniform(0.5, 1.5, B)
        pv = rs.uniform(-1, 1, (B, 2)); self_s[:, 6:8] = pv / np.maximum(np.linalg.norm(pv, axis=1, keepdims=True), 1.0)
        others = np.zeros((B, n_other, 5), np.float32)
        others[:, :, 0:2] = rs.uniform(-spread, spread, (B, n_other, 2))   # dense scenes overlap -> exercises LP3
        others[:, :, 2:4] = rs.uniform(-1, 1, (B, n_other, 2))
        others[:, :, 4] = rs.uniform(0.46, 0.66, (B, n_other))
        nd = 4.0 if spread > 3 else 10.0
        out = orca_solve(torch.from_numpy(self_s).cuda(), torch.from_numpy(others).cuda(), neighbor_dist=nd).cpu().numpy()
        n_lp3 = 0
        for b in range(B):
            (vx, vy), lines, fail = O.orca_velocity(self_s[b], others[b], neighbor_dist=nd, want_lines=True)
            n_lp3 += int(fail < len(lines))
            assert out[b, 0] == np.float32(vx) and out[b, 1] == np.float32(vy), (n_other, spread, b, out[b], (vx, vy))
        if spread <= 1.5 and n_other >= 19:
            assert n_lp3 > 0  # the infeasible (LP3) branch was really exercised
