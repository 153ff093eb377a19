"""Pin the CPU oracle against golden vectors captured from the Python reference.

Bar: flags / indices / sort order exact; float32 observations equal to <= 1e-6 (the oracle's sin/cos is its own
deterministic <=1ulp(fp64) implementation, numpy's is libm -- see DESIGN.md); fp64 state <= 1e-9.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import golden_util as G


def test_mt19937_matches_numpy_legacy_stream():
    for seed in (0, 425, 2425, 2 ** 32 - 1):
        np.random.seed(seed)
        ref = [np.random.random() for _ in range(1500)]  # crosses two twist boundaries
        m = O.MT(seed)
        assert ref == [m.random() for _ in range(1500)]


def test_randint_matches_numpy_legacy_stream():
    """RandomState.randint (crowd size draws, crowd_sim_var_num.py:103,423): masked rejection on 32-bit words, no draw for a
    single-value range -- values AND stream position must agree."""
    for seed in (1, 425):
        np.random.seed(seed)
        m = O.MT(seed)
        for lo, hi in [(0, 1), (0, 2), (0, 3), (3, 8), (15, 26), (1, 21), (0, 6), (0, 1000), (5, 6)] * 40:
            assert np.random.randint(lo, hi) == m.randint(lo, hi)
            assert np.random.random() == m.random()


def test_normal_matches_numpy_legacy_stream():
    """RandomState.normal (the wheel dead-band noise of smooth_action, crowd_sim.py:338-350): polar Box-Muller on the same MT19937 stream with
    the cached second deviate.  The uniform draws it consumes are bit-identical, so the STREAM POSITION must agree exactly (checked through the
    next uniform); the values agree to the last bit or two (the oracle's deterministic log is <= 1 ulp from libm)."""
    for seed in (3, 425, 1425):
        np.random.seed(seed)
        m = O.MT(seed)
        worst = 0.0
        for i in range(400):
            a, b = np.random.normal(1.8, 0.15), m.normal(1.8, 0.15)
            worst = max(worst, abs(a - b))
            if i % 7 == 0:   # interleave uniforms: an odd number of normals leaves a cached deviate behind, which must survive them
                assert np.random.random() == m.random()
        assert worst <= 1e-15
        assert np.random.random() == m.random()


def test_field_of_view_test_is_the_reference_decision():
    """detect_visible's cone test (crowd_sim.py:513-537): the oracle (and the kernel) evaluate clip(v_fov . v_12) >= cos(fov / 2) on
    v_fov = v / |v|; the reference evaluates arccos(clip(v_fov . v_12)) <= fov / 2 on v_fov = (cos, sin)(arctan2(vy, vx)).  Same decision on
    random pairs, on agents at rest (heading +x; -x for vx = -0.0), behind / ahead exactly, and for the unicycle form (heading = theta)."""
    rs = np.random.RandomState(5)

    def ref(unicycle, fov_pi, p1, v1, th1, p2):
        real_theta = th1 if unicycle else np.arctan2(v1[1], v1[0])
        v_fov = np.array([np.cos(real_theta), np.sin(real_theta)])
        v_12 = np.array([p2[0] - p1[0], p2[1] - p1[1]])
        with np.errstate(invalid="ignore", divide="ignore"):
            v_fov = v_fov / np.linalg.norm(v_fov)
            v_12 = v_12 / np.linalg.norm(v_12)
            offset = np.arccos(np.clip(np.dot(v_fov, v_12), a_min=-1, a_max=1))
        return bool(np.abs(offset) <= np.pi * fov_pi / 2)

    cases = []
    for _ in range(20000):
        cases.append((int(rs.rand() < 0.3), float(rs.choice([0.5, 0.8, 1.0, 1.2, 1.5, 1.99, 2.0])), rs.uniform(-8, 8, 2), rs.uniform(-1.2, 1.2, 2),
                      float(rs.uniform(0, 2 * np.pi)), rs.uniform(-8, 8, 2)))
    z = np.zeros(2)
    cases += [(0, 1.0, z, z, 0.0, np.array([1.0, 0.0])), (0, 1.0, z, z, 0.0, np.array([-1.0, 0.0])), (0, 1.0, z, np.array([-0.0, 0.0]), 0.0, np.array([-1.0, 0.0])),
              (0, 1.0, z, np.array([-0.0, 0.0]), 0.0, np.array([1.0, 0.0])), (0, 2.0, z, np.array([1.0, 0.0]), 0.0, np.array([-3.0, 0.0])),
              (0, 1.5, z, np.array([0.3, 0.4]), 0.0, z), (1, 1.0, z, z, np.pi / 2, np.array([0.0, 2.0])), (1, 1.0, z, z, np.pi / 2, np.array([0.0, -2.0]))]
    for u, fov, p1, v1, th, p2 in cases:
        got = bool(O.lib().orc_in_fov(u, fov, float(p1[0]), float(p1[1]), float(v1[0]), float(v1[1]), th, float(p2[0]), float(p2[1])))
        assert got == ref(u, fov, p1, v1, th, p2), (u, fov, p1, v1, th, p2)


def test_log_accuracy():
    """The deterministic log that stands in for libm's inside the normal sampler: within 1 ulp over (0, 1] and across the binades."""
    rs = np.random.RandomState(1)
    xs = np.concatenate([rs.uniform(0, 1, 100000), 10.0 ** rs.uniform(-300, 0, 10000), 1 - 10.0 ** rs.uniform(-16, -1, 10000),
                         [0.5, 0.25, 1.0, 0.7071067811865476, 0.9999999999999999]])
    xs = xs[xs > 2.3e-308]
    got = np.array([O.lib().orc_log(float(x)) for x in xs])
    want = np.log(xs)
    assert np.max(np.abs(got - want) / np.spacing(np.abs(want) + 1e-320)) <= 1.0
    assert O.lib().orc_log(1.0) == 0.0


def test_sincos_accuracy():
    xs = np.linspace(-0.07, 2 * np.pi + 0.07, 20001)  # the unicycle heading + one clipped rotation leaves [0, 2 pi) by <= 0.06
    got = np.array([O.sincos(x) for x in xs])
    assert np.max(np.abs(got[:, 0] - np.sin(xs))) <= 2.3e-16
    assert np.max(np.abs(got[:, 1] - np.cos(xs))) <= 2.3e-16


def test_exp_accuracy():
    """The deterministic exp that stands in for np.exp in the social-force policy: within 1 ulp of libm over the arguments it sees."""
    xs = np.concatenate([np.linspace(-40.0, 3.0, 40001), np.linspace(-1e-3, 1e-3, 2001), [0.0, -700.5, 12.25]])
    got = np.array([O.lib().orc_exp(float(x)) for x in xs])
    want = np.exp(xs)
    ok = xs >= -700.0   # below that the stand-in flushes to zero (never reached: the argument is (r_i + r_j - d) / B)
    assert np.max(np.abs(got[ok] - want[ok]) / np.spacing(want[ok])) <= 1.0
    assert got[xs == 0.0][0] == 1.0 and got[xs == -700.5][0] == 0.0


@pytest.mark.parametrize("path", G.env_fixtures(), ids=lambda p: p.split("env_")[-1][:-4])
def test_env_trace_matches_reference(path):
    z, meta = G.load(path)
    cfg = O.default_config(**G.sim_kwargs(meta, oracle=True))
    env = O.OracleEnv(cfg, meta["seed"] + meta["rank"])
    ob = env.reset()
    for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num"):
        np.testing.assert_allclose(ob[k], z["reset_" + k], rtol=0, atol=1e-6, err_msg="reset " + k)
    has_masks = meta["env_name"] != "CrowdSimPred-v0"  # CrowdSimPred's obs dict has no visible_masks key
    if has_masks:
        np.testing.assert_array_equal(ob["visible_masks"], z["reset_visible_masks"])
    T = len(z["done"])
    n_bit_equal = n_vals = 0
    for t in range(T):
        ob, r, done, info = env.step(z["actions"][t], autoreset=True)
        assert done == bool(z["done"][t]), "done @%d" % t
        assert info["info"] == int(z["info"][t]), "info @%d" % t
        assert np.float32(r) == pytest.approx(z["reward"][t], abs=1e-6), "reward @%d" % t
        if "min_dist" in z.files:  # test-phase traces: Danger(min_dist) from the humans' true future positions
            assert info["min_dist"] == pytest.approx(float(z["min_dist"][t]), abs=1e-9), "min_dist @%d" % t
        if "human_count" in z.files:  # crowd size after this step (after the auto-reset when the episode ended)
            assert env.human_count == int(z["human_count"][t]), "human_count @%d" % t
        if done:
            assert info["episode"]["l"] == int(z["ep_len"][t])
            assert info["episode"]["r"] == pytest.approx(float(z["ep_return"][t]), abs=2e-6)
        for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num"):
            np.testing.assert_allclose(ob[k], z[k][t], rtol=0, atol=1e-6, err_msg="%s @%d" % (k, t))
            n_bit_equal += int(np.sum(ob[k] == z[k][t]))
            n_vals += ob[k].size
        if has_masks:
            np.testing.assert_array_equal(ob["visible_masks"], z["visible_masks"][t], err_msg="visible_masks @%d" % t)
    # float32 observations are bit-identical except where a 1ulp(fp64) sin/cos difference crosses a rounding boundary
    assert n_bit_equal / n_vals > 0.999


COLLECT = sorted(__import__("glob").glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collect_*.npz")))


@pytest.mark.parametrize("path", COLLECT, ids=lambda p: os.path.basename(p)[8:-4])
def test_collect_env_trace_matches_reference_exactly(path):
    """CrowdSimVarNumCollect-v0 (crowd_sim_var_num_collect.py, stepped like collect_data.py does): every pred_info entry -- frame ids,
    prediction ids (fresh id after a human left the robot's view), float32 positions, the inf pattern --, the info codes and the
    robot's re-drawn goals (median of the humans / uniform point) equal the reference's, 600 steps."""
    import json
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    over = meta["over"]
    cfg = O.default_config(human_num=int(over["sim.human_num"]), env_kind=O.ENV_COLLECT, robot_policy=1, nenv=meta["nenv"], phase=0,
                           randomize_attributes=int(bool(over["env.randomize_attributes"])),
                           random_goal_changing=int(bool(over["humans.random_goal_changing"])),
                           end_goal_changing=int(bool(over["humans.end_goal_changing"])))
    env = O.OracleEnv(cfg, meta["seed"] + meta["rank"])
    np.testing.assert_array_equal(env.reset()["spatial_edges"], z["reset_pred_info"])
    goal_changes = 0
    for t in range(len(z["info"])):
        ob, r, d, inf = env.step(np.zeros(2, np.float32))
        np.testing.assert_array_equal(ob["spatial_edges"], z["pred_info"][t], err_msg="pred_info @%d" % t)
        assert inf["info"] == int(z["info"][t]) and r == 0.0 and not d
        rn, rs = ob["robot_node"].ravel(), z["robot_state"][t]
        assert rn[3] == np.float32(rs[5]) and rn[4] == np.float32(rs[6]), "goal @%d" % t
        goal_changes += int(inf["info"] == 3)
    assert goal_changes >= 4
