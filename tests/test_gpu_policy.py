"""-m gpu: the HIP policy forward and rollout math against golden outputs of the reference and the numpy oracle."""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from tests import policy_util as PU  # noqa: E402
from tests.golden_util import GOLDEN  # noqa: E402

TOL = 1e-4  # north_star: attention outputs / returns within 1e-4 fp32


def _sd_dev(shapes):
    sd = PU.formula_state_dict({k: tuple(v) for k, v in shapes.items()})
    return sd, {k: torch.from_numpy(v).cuda() for k, v in sd.items()}


def _dev(obs):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in obs.items()}


@pytest.mark.parametrize("mode", ["fused", "bf16x3", "fp32"])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "policy_*_h*.npz"))), ids=lambda p: os.path.basename(p)[7:-4])
def test_policy_act_matches_reference_golden(path, mode):
    from crowdnav_prediction_attngraph_amd.hip import HipPolicy
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    E, H, D = meta["E"], meta["H"], meta["D"]
    _, sd = _sd_dev(meta["shapes"])
    pol = HipPolicy(H, D, max(E, 4))
    pol.set_gemm_mode(mode)     # bf16x3 = split-precision MFMA (default), fp32 = exact fp32 MFMA: both must hold 1e-4
    pol.set_weights(sd)
    obs = _dev({k: z[k] for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")})
    hxs, masks = torch.from_numpy(z["hxs_node"]).cuda(), torch.from_numpy(z["masks"]).cuda()
    out = pol.act(obs, hxs, masks, eps=None)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out["value"].cpu().numpy(), z["value"], atol=TOL)
    np.testing.assert_allclose(out["action"].cpu().numpy(), z["action"], atol=TOL)
    np.testing.assert_allclose(out["logp"].cpu().numpy(), z["logp"], atol=TOL)
    np.testing.assert_allclose(out["hxs"].cpu().numpy(), z["hx_out"], atol=TOL)
    taps = pol.taps(E)
    det = z["detected_human_num"].reshape(E).astype(int)
    valid = np.arange(H)[None, :] < det[:, None]
    sl = taps["spatial_lin"].cpu().numpy()
    np.testing.assert_allclose(sl[valid], z["spatial_lin"][valid], atol=TOL)   # padded rows are not materialised
    np.testing.assert_allclose(taps["hr_attn"].cpu().numpy(), z["hr_attn"], atol=TOL)
    np.testing.assert_allclose(taps["hr_out"].cpu().numpy(), z["hr_out"], atol=TOL)
    np.testing.assert_allclose(taps["robot_emb"].cpu().numpy(), z["robot_emb"], atol=TOL)
    np.testing.assert_allclose(taps["actor_feat"].cpu().numpy(), z["actor_feat"], atol=TOL)
    # stochastic path: action = mean + std * eps and its log-prob, checked through the fixed-action golden
    mean = z["action"]
    std = np.exp(PU.formula_state_dict({k: tuple(v) for k, v in meta["shapes"].items()})["dist.logstd._bias"].reshape(1, 2))
    eps = ((z["fixed_action"] - mean) / std).astype(np.float32)
    out2 = pol.act(obs, hxs, masks, eps=torch.from_numpy(eps).cuda())
    np.testing.assert_allclose(out2["action"].cpu().numpy(), z["fixed_action"], atol=TOL)
    np.testing.assert_allclose(out2["logp"].cpu().numpy(), z["logp_fixed"], atol=TOL)
    v = pol.get_value(obs, hxs, masks)
    np.testing.assert_allclose(v.cpu().numpy(), z["value"], atol=TOL)


@pytest.mark.parametrize("mode", ["fused", "bf16x3", "fp32"])
@pytest.mark.parametrize("E,H,D", [(257, 20, 2), (130, 20, 12), (64, 5, 2), (40, 50, 2), (21, 57, 2), (9, 63, 12), (3, 64, 2), (1, 1, 2)])
def test_policy_act_matches_numpy_oracle(E, H, D, mode):
    """Sizes with ragged tiles (E*H not a multiple of 128), random-looking weights, random detected counts."""
    from crowdnav_prediction_attngraph_amd.hip import HipPolicy
    from oracle import policy_oracle as P
    shapes = json.loads(str(np.load(os.path.join(GOLDEN, "policy_varnum_e4_h20.npz"))["meta"]))["shapes"]
    shapes["base.spatial_attn.embedding_layer.0.weight"] = [128, D]
    sd, sdd = _sd_dev(shapes)
    pol = HipPolicy(H, D, E)
    pol.set_gemm_mode(mode)
    pol.set_weights(sdd)
    obs = PU.synth_obs(E, H, D, seed=E + H)
    rs = np.random.RandomState(1)
    hxs = rs.uniform(-1, 1, (E, 1, 128)).astype(np.float32)
    masks = (rs.uniform(size=(E, 1)) > 0.2).astype(np.float32)
    eps = rs.standard_normal((E, 2)).astype(np.float32)
    out = pol.act(_dev(obs), torch.from_numpy(hxs).cuda(), torch.from_numpy(masks).cuda(), eps=torch.from_numpy(eps).cuda())
    value, mean, _, h_new, feat = P.act(sd, obs, hxs.reshape(E, 128), masks)
    std = np.exp(sd["dist.logstd._bias"].astype(np.float64).reshape(1, 2))
    action = mean + std * eps
    logp = P.log_prob(mean, np.log(std), action)
    np.testing.assert_allclose(out["value"].cpu().numpy(), value, atol=TOL)
    np.testing.assert_allclose(out["action"].cpu().numpy(), action, atol=TOL)
    np.testing.assert_allclose(out["logp"].cpu().numpy(), logp, atol=TOL)
    np.testing.assert_allclose(out["hxs"].cpu().numpy().reshape(E, 128), h_new, atol=TOL)


def test_gae_and_advantage_norm_match_reference_golden():
    from crowdnav_prediction_attngraph_amd import hip
    from oracle import oracle as O
    for path in sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz"))):
        z = np.load(path)
        values = np.concatenate([z["values"], z["next_value"][None]], 0).astype(np.float32)
        T, N = z["rewards"].shape[:2]
        ret = torch.zeros(T + 1, N, 1, device="cuda")
        hip.gae(torch.from_numpy(z["rewards"]).cuda(), torch.from_numpy(values).cuda(), torch.from_numpy(z["masks"]).cuda(), 0.99, 0.95, ret)
        np.testing.assert_allclose(ret[:-1].cpu().numpy(), z["returns"][:-1], atol=1e-5)
        # bit-exact against the C oracle (same fp32 op order)
        o = O.gae(z["rewards"].reshape(T, N), values.reshape(T + 1, N), z["masks"].reshape(T + 1, N), 0.99, 0.95)
        assert np.array_equal(ret[:-1].cpu().numpy().reshape(T, N), o)
        vals_d = torch.from_numpy(values).cuda()
        stats = hip.adv_stats(ret, vals_d, T * N)
        adv = torch.zeros(T, N, 1, device="cuda")
        hip.adv_normalize(ret, vals_d, stats, T * N, adv)
        np.testing.assert_allclose(adv.cpu().numpy(), z["adv_norm"], atol=1e-4)


def test_gae_large_matches_oracle_and_properties():
    from crowdnav_prediction_attngraph_amd import hip
    from oracle import oracle as O
    rs = np.random.RandomState(0)
    T, N = 30, 4096
    r = rs.uniform(-1, 1, (T, N)).astype(np.float32)
    v = rs.uniform(-2, 2, (T + 1, N)).astype(np.float32)
    m = (rs.uniform(size=(T + 1, N)) > 0.03).astype(np.float32)
    ret = torch.zeros(T + 1, N, 1, device="cuda")
    hip.gae(torch.from_numpy(r).cuda().view(T, N, 1), torch.from_numpy(v).cuda().view(T + 1, N, 1), torch.from_numpy(m).cuda().view(T + 1, N, 1), 0.99, 0.95, ret)
    got = ret[:-1].cpu().numpy().reshape(T, N)
    assert np.array_equal(got, O.gae(r, v, m, 0.99, 0.95))
    # property: with lambda = 1 and no terminations the return is the discounted reward sum + bootstrapped value
    m1 = np.ones_like(m)
    ret1 = torch.zeros(T + 1, N, 1, device="cuda")
    hip.gae(torch.from_numpy(r).cuda().view(T, N, 1), torch.from_numpy(v).cuda().view(T + 1, N, 1), torch.from_numpy(m1).cuda().view(T + 1, N, 1), 0.99, 1.0, ret1)
    disc = v[T].astype(np.float64)
    for t in reversed(range(T)):
        disc = r[t] + 0.99 * disc
    np.testing.assert_allclose(ret1[0].cpu().numpy().reshape(N), disc, atol=2e-4)
    # normalised advantages: zero mean, unit (unbiased) std
    vals_d = torch.from_numpy(v).cuda().view(T + 1, N, 1)
    stats = hip.adv_stats(ret, vals_d, T * N)
    adv = torch.zeros(T, N, 1, device="cuda")
    hip.adv_normalize(ret, vals_d, stats, T * N, adv)
    a = adv.cpu().numpy().astype(np.float64)
    assert abs(a.mean()) < 1e-5 and abs(a.std(ddof=1) - 1.0) < 1e-4
