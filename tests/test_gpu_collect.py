"""-m gpu: CrowdSimVarNumCollect-v0 (the GST dataset generator of collect_data.py) on the device: bit-exact against the oracle, exact
against the reference's own traces, and the text files collect_data.py writes."""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "collect_*.npz")))


def _kw(meta):
    over = meta["over"]
    return dict(human_num=int(over["sim.human_num"]), env_kind=3, robot_policy=1, nenv=meta["nenv"], phase=0,
                randomize_attributes=int(bool(over["env.randomize_attributes"])), random_goal_changing=int(bool(over["humans.random_goal_changing"])),
                end_goal_changing=int(bool(over["humans.end_goal_changing"])))


@pytest.mark.parametrize("kw", [dict(human_num=20), dict(human_num=10, randomize_attributes=1, random_goal_changing=1),
                                dict(human_num=7, robot_visible=1), dict(human_num=33, circle_radius=9.0)],
                         ids=["h20", "h10_rand", "h7_robotvisible", "h33"])
def test_hip_collect_env_matches_oracle_bit_exact(kw):
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
    from oracle import oracle as O
    E, T, seed = 32, 900, 425
    kw = dict(kw, env_kind=3, robot_policy=1, nenv=E)
    env = HipEnvBatch(A.default_env_config(**kw), E, seed)
    oenvs = [O.OracleEnv(O.default_config(**kw), seed + i) for i in range(E)]
    obs = env.reset()
    assert obs["spatial_edges"].shape == (E, kw["human_num"], 4)
    pred = obs["spatial_edges"].cpu().numpy()
    for i, oe in enumerate(oenvs):
        np.testing.assert_array_equal(pred[i], oe.reset()["spatial_edges"])
    act = torch.zeros(E, 2, device=env.device)
    zero = np.zeros(2, np.float32)
    infos, max_id = set(), 0
    for t in range(T):
        obs, rew, done, info, _, _ = env.step(act)
        pred, info_h, rn = obs["spatial_edges"].cpu().numpy(), info.cpu().numpy(), obs["robot_node"].cpu().numpy()
        assert not done.any() and float(rew.abs().max()) == 0.0
        for i, oe in enumerate(oenvs):
            ob, r, d, inf = oe.step(zero)
            np.testing.assert_array_equal(pred[i], ob["spatial_edges"], err_msg="pred_info t=%d env=%d" % (t, i))   # inf == inf
            np.testing.assert_array_equal(rn[i].reshape(1, 7), ob["robot_node"], err_msg="robot_node t=%d env=%d" % (t, i))
            assert int(info_h[i]) == inf["info"], (t, i)
            infos.add(inf["info"])
        max_id = max(max_id, int(pred[:, :, 1].max()))
    assert 3 in infos                      # robots reached goals and drew new ones (median / uniform)
    assert max_id > 3 * kw["human_num"]    # humans left the robot's view and came back under fresh ids
    env.close()


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[8:-4])
def test_hip_collect_env_replays_reference_traces_and_file_lines(path):
    """The reference's CrowdSimVarNumCollect env stepped by collect_data.py's loop (tests/golden/make_golden_collect.py): pred_info of
    every step exact, and the batched collector's text lines for that env equal to the lines the reference script would write."""
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.collect import CollectVecEnv, collect_lines, format_rows
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    over = dict(meta["over"])
    over.pop("sim.predict_method", None); over.pop("env.use_wrapper", None)
    cfg = C.Config(**dict(over, **{"robot.policy": "orca"}))
    envs = CollectVecEnv(meta["seed"], meta["nenv"], torch.device("cuda", 0), config=cfg)
    r = meta["rank"]
    ob = envs.reset()["pred_info"]
    np.testing.assert_array_equal(ob[r], z["reset_pred_info"])
    lines = format_rows(ob[r])
    T = len(z["info"])
    for t in range(T):
        ob, rew, done, infos = envs.step(np.zeros((meta["nenv"], 2)))
        np.testing.assert_array_equal(ob["pred_info"][r], z["pred_info"][t], err_msg="pred_info @%d" % t)
        assert type(infos[r]["info"]).__name__ == {0: "Nothing", 1: "Timeout", 2: "Collision", 3: "ReachGoal"}[int(z["info"][t])]
        assert rew[r] == 0 and not done[r]
        if t + 1 < T:
            lines += format_rows(ob["pred_info"][r])
    assert lines == str(z["lines"]).split("\n")
    envs.close()
    # the block-buffered collector (observations stay on the device, one transfer per block) gives the same lines
    envs = CollectVecEnv(meta["seed"], meta["nenv"], torch.device("cuda", 0), config=cfg)
    got = collect_lines(envs, T, 1, block=64)
    assert got[r] == str(z["lines"]).split("\n")
    envs.close()


def test_collect_data_writes_one_file_per_env(tmp_path):
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.collect import collectData
    cfg = C.non_randomized(**{"sim.human_num": 20, "data.tot_steps": 120, "data.data_save_dir": str(tmp_path / "ds")})
    out = collectData(torch.device("cuda", 0), True, cfg, num_envs=64, seed=425)
    files = sorted(os.listdir(out), key=lambda s: int(s[:-4]))
    assert out.endswith("train") and files == ["%d.txt" % i for i in range(64)]
    rows = np.loadtxt(os.path.join(out, "0.txt"))
    assert rows.shape[1] == 4 and rows[:, 0].min() == 0 and rows[:, 0].max() == 119 and np.isfinite(rows).all()
    z = np.load(os.path.join(GOLDEN, "collect_h20_nonrand_r0.npz"))          # same seed, same config: env 0 is the fixture's env
    assert open(os.path.join(out, "0.txt")).read().split("\n")[:200] == str(z["lines"]).split("\n")[:200]
