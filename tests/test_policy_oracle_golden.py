"""Pin the numpy policy/rollout oracle against golden outputs of the reference's torch code."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import policy_oracle as P
from tests import policy_util as PU
from tests.golden_util import GOLDEN

TOL = 2e-5  # reference is torch fp32; north_star tolerance is 1e-4


def _sd(meta):
    return PU.formula_state_dict({k: tuple(v) for k, v in meta["shapes"].items()})


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "policy_*_h*.npz"))), ids=lambda p: os.path.basename(p)[7:-4])
def test_policy_act_matches_reference(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    sd = _sd(meta)
    E = meta["E"]
    obs = {k: z[k] for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
    taps = {}
    value, action, logp, h_new, feat = P.act(sd, obs, z["hxs_node"].reshape(E, 128), z["masks"], taps=taps)
    np.testing.assert_allclose(value, z["value"], atol=TOL)
    np.testing.assert_allclose(action, z["action"], atol=TOL)
    np.testing.assert_allclose(logp, z["logp"], atol=TOL)
    np.testing.assert_allclose(h_new, z["hx_out"].reshape(E, 128), atol=TOL)
    np.testing.assert_allclose(feat, z["actor_feat"], atol=TOL)
    for k in ("hh_out", "spatial_lin", "hr_out", "hr_attn", "robot_emb"):
        np.testing.assert_allclose(taps[k], z[k], atol=5e-5, err_msg=k)
    _, _, lpf, _, _ = P.act(sd, obs, z["hxs_node"].reshape(E, 128), z["masks"], action=z["fixed_action"].astype(np.float64))
    np.testing.assert_allclose(lpf, z["logp_fixed"], atol=TOL)
    assert P.entropy_mean(sd["dist.logstd._bias"].astype(np.float64).reshape(1, -1)) == pytest.approx(float(z["entropy"]), abs=1e-6)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz"))), ids=lambda p: os.path.basename(p)[8:-4])
def test_rollout_math_matches_reference(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    sd = _sd(meta)
    T, E, nmb = meta["T"], meta["E"], meta["nmb"]
    values = np.concatenate([z["values"], z["next_value"][None]], 0)
    ret = P.gae(z["rewards"].astype(np.float64), values.astype(np.float64), z["masks"].astype(np.float64), 0.99, 0.95)
    np.testing.assert_allclose(ret, z["returns"][:-1], atol=TOL)
    np.testing.assert_allclose(P.adv_normalize(ret, values[:-1].astype(np.float64)), z["adv_norm"], atol=1e-4)
    # evaluate_actions over the first minibatch layout
    N = E // nmb
    obs_seq = [{k: z["obs%d_%s" % (s, k)][:N] for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")} for s in range(T)]
    ev_v, ev_lp, ev_ent = P.evaluate_actions(sd, obs_seq, z["hxs_node"][0, :N].reshape(N, 128), z["masks"][:-1, :N], z["actions"][:, :N])
    np.testing.assert_allclose(ev_v, z["ev_values"], atol=TOL)
    np.testing.assert_allclose(ev_lp, z["ev_logp"], atol=TOL)
    assert ev_ent == pytest.approx(float(z["ev_entropy"]), abs=1e-6)
    # the act-time values/log-probs stored in the rollout are consistent with a sequential re-evaluation
    np.testing.assert_allclose(ev_v.reshape(T, N, 1), z["values"][:, :N], atol=TOL)
    np.testing.assert_allclose(ev_lp.reshape(T, N, 1), z["logp"][:, :N], atol=TOL)
