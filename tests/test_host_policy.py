"""CPU tests of the host-side mirror (Policy / RolloutStorage / PPO) against golden outputs of the reference."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
from crowdnav_prediction_attngraph_amd.ppo import PPO
from crowdnav_prediction_attngraph_amd.storage import RolloutStorage
from tests import policy_util as PU
from tests.golden_util import GOLDEN


def _policy(meta, E, nmb=1, T=1):
    ob_space, act_space = make_spaces(meta["H"], meta["D"])
    pol = Policy(ob_space.spaces, act_space, base="selfAttn_merge_srnn",
                 base_kwargs=dict(env_name=meta["env_name"], num_processes=E, num_mini_batch=nmb, seq_length=T))
    return pol, ob_space, act_space


def _load_formula(pol, meta):
    sd = PU.formula_state_dict({k: tuple(v) for k, v in meta["shapes"].items()})
    pol.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})


def test_state_dict_keys_shapes_and_order_match_reference():
    z = np.load(os.path.join(GOLDEN, "policy_varnum_e4_h20.npz"))
    meta = json.loads(str(z["meta"]))
    pol, _, _ = _policy(meta, 4)
    got = [(k, list(v.shape)) for k, v in pol.state_dict().items()]
    assert got == [(k, list(v)) for k, v in meta["shapes"].items()]
    assert sum(p.numel() for p in pol.parameters()) == 2501255  # SURVEY.md 3.3 [probed]


@pytest.mark.parametrize("tag,env_name,H,D", [("varnum_h20", "CrowdSimVarNum-v0", 20, 2), ("pred_h20", "CrowdSimPred-v0", 20, 12)])
def test_seeded_init_matches_reference(tag, env_name, H, D):
    """Same construction order -> same RNG consumption -> the same initial weights as the reference under one seed.
    Uniform-initialised tensors are bit-identical; orthogonal_ ones go through LAPACK QR, whose last bits depend on the
    BLAS thread count, hence the 1e-6 tolerance."""
    ref = np.load(os.path.join(GOLDEN, "policy_init.npz"))
    torch.manual_seed(0)
    pol, _, _ = _policy(dict(H=H, D=D, env_name=env_name), 16, 2, 30)
    for k, v in pol.state_dict().items():
        a = v.detach().numpy().astype(np.float64)
        got = np.array([a.sum(), np.abs(a).sum(), float(a.ravel()[0]), float(a.ravel()[-1])])
        np.testing.assert_allclose(got, ref["%s/%s" % (tag, k)], rtol=1e-6, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "policy_*_h*.npz"))), ids=lambda p: os.path.basename(p)[7:-4])
def test_cpu_act_matches_reference_golden(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    E = meta["E"]
    pol, _, _ = _policy(meta, E)
    _load_formula(pol, meta)
    obs = {k: torch.from_numpy(z[k]) for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
    hxs = {"human_node_rnn": torch.from_numpy(z["hxs_node"]), "human_human_edge_rnn": torch.zeros(E, meta["H"] + 1, 256)}
    value, action, logp, hx = pol.act(obs, hxs, torch.from_numpy(z["masks"]), deterministic=True)
    np.testing.assert_allclose(value.numpy(), z["value"], atol=2e-5)
    np.testing.assert_allclose(action.numpy(), z["action"], atol=2e-5)
    np.testing.assert_allclose(logp.numpy(), z["logp"], atol=2e-5)
    np.testing.assert_allclose(hx["human_node_rnn"].numpy(), z["hx_out"], atol=2e-5)
    assert hx["human_human_edge_rnn"].shape == (E, meta["H"] + 1, 256) and float(hx["human_human_edge_rnn"].abs().sum()) == 0.0
    np.testing.assert_allclose(pol.get_value(obs, hxs, torch.from_numpy(z["masks"])).numpy(), z["value"], atol=2e-5)


def _fill_rollouts(z, meta, pol, ob_space, act_space):
    T, E, H = meta["T"], meta["E"], meta["H"]
    ro = RolloutStorage(T, E, ob_space.spaces, act_space, 128, 256)
    keys = ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")
    for k in keys:
        ro.obs[k][0].copy_(torch.from_numpy(z["obs0_" + k]))
    for s in range(T):
        nxt = {k: torch.from_numpy(z["obs%d_%s" % (s + 1, k)]) for k in keys}
        nxt["visible_masks"] = torch.zeros(E, H, dtype=torch.bool)
        hx = {"human_node_rnn": torch.from_numpy(z["hxs_node"][s + 1]), "human_human_edge_rnn": torch.zeros(1, 1, 1).expand(E, H + 1, 256)}
        masks = torch.from_numpy(np.where(z["dones"][s], 0.0, 1.0).astype(np.float32).reshape(E, 1))
        ro.insert(nxt, hx, torch.from_numpy(z["actions"][s]), torch.from_numpy(z["logp"][s]), torch.from_numpy(z["values"][s]),
                  torch.from_numpy(z["rewards"][s]), masks, torch.ones(E, 1))
    return ro


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz"))), ids=lambda p: os.path.basename(p)[8:-4])
def test_storage_returns_evaluate_and_ppo_update_match_reference(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    T, E, nmb = meta["T"], meta["E"], meta["nmb"]
    pol, ob_space, act_space = _policy(meta, E, nmb, T)
    _load_formula(pol, meta)
    ro = _fill_rollouts(z, meta, pol, ob_space, act_space)
    np.testing.assert_array_equal(ro.masks.numpy(), z["masks"])
    ro.compute_returns(torch.from_numpy(z["next_value"]), True, 0.99, 0.95, False)
    np.testing.assert_allclose(ro.returns.numpy()[:-1], z["returns"][:-1], atol=1e-6)
    # evaluate_actions on the first-minibatch layout (envs 0..N-1 in order)
    N = E // nmb
    flat = lambda x: x[:T, :N].reshape(T * N, *x.shape[2:])  # noqa: E731
    with torch.no_grad():
        ob_b = {k: flat(v) for k, v in ro.obs.items()}
        hx_b = {"human_node_rnn": ro.recurrent_hidden_states["human_node_rnn"][0, :N], "human_human_edge_rnn": None}
        v, lp, ent, _ = pol.evaluate_actions(ob_b, hx_b, flat(ro.masks), flat(ro.actions))
    np.testing.assert_allclose(v.numpy(), z["ev_values"], atol=2e-5)
    np.testing.assert_allclose(lp.numpy(), z["ev_logp"], atol=2e-5)
    assert float(ent) == pytest.approx(float(z["ev_entropy"]), abs=1e-6)
    # one full update(): same torch.randperm draws as the reference (CPU generator, seed 321)
    agent = PPO(pol, 0.2, meta["ppo_epoch"], nmb, 0.5, 0.0, lr=4e-5, eps=1e-5, max_grad_norm=0.5)
    torch.manual_seed(meta["update_seed"])
    v_loss, a_loss, ent = agent.update(ro)
    np.testing.assert_allclose([v_loss, a_loss, ent], z["losses"], atol=2e-6)
    sd = pol.state_dict()
    for k in ("dist.fc_mean.weight", "base.critic_linear.weight", "base.robot_linear.0.weight"):
        np.testing.assert_allclose(sd[k].numpy(), z["after_" + k], atol=2e-6, err_msg=k)
    for k, t in sd.items():
        a = t.numpy().astype(np.float64)
        np.testing.assert_allclose([a.sum(), np.abs(a).sum()], z["chk_" + k], rtol=2e-6, atol=2e-4, err_msg=k)
    ro.after_update()
    assert torch.equal(ro.obs["spatial_edges"][0], ro.obs["spatial_edges"][-1]) and torch.equal(ro.masks[0], ro.masks[-1])


def test_recurrent_generator_layout_matches_reference_semantics():
    """T-major flattening of env trajectories, hidden state only at t = 0, a permutation of all envs per epoch."""
    T, E, H, D = 3, 6, 4, 2
    ob_space, act_space = make_spaces(H, D)
    ro = RolloutStorage(T, E, ob_space.spaces, act_space, 128, 256)
    ro.actions.copy_(torch.arange(T * E * 2, dtype=torch.float32).view(T, E, 2))
    ro.recurrent_hidden_states["human_node_rnn"][0, :, 0, 0] = torch.arange(E, dtype=torch.float32)
    adv = torch.arange(T * E, dtype=torch.float32).view(T, E, 1)
    torch.manual_seed(5)
    perm = torch.randperm(E)
    torch.manual_seed(5)
    seen = []
    for b, (obs_b, hx_b, act_b, vp_b, ret_b, m_b, lp_b, adv_b) in enumerate(ro.recurrent_generator(adv, 2)):
        idx = perm[b * 3:(b + 1) * 3]
        seen += idx.tolist()
        assert act_b.shape == (T * 3, 2) and obs_b["spatial_edges"].shape == (T * 3, H, D)
        assert torch.equal(act_b.view(T, 3, 2), ro.actions[:, idx])
        assert torch.equal(adv_b.view(T, 3, 1), adv[:, idx])
        assert torch.equal(hx_b["human_node_rnn"][:, 0, 0], idx.float())
        assert hx_b["human_human_edge_rnn"].shape == (3, H + 1, 256)
    assert sorted(seen) == list(range(E))
    with pytest.raises(AssertionError):
        list(ro.recurrent_generator(adv, E + 1))
    # the group size is num_processes // num_mini_batch and range(0, num_processes, size) decides how many groups there are
    # (storage.py:190-192): 6 envs with 4 mini-batches -> 6 groups of one env; an incomplete last group raises IndexError only
    # after the complete ones were yielded (storage.py:209-210)
    assert [b[2].shape[0] for b in ro.recurrent_generator(adv, 4)] == [T] * 6
    ro8 = RolloutStorage(T, 8, ob_space.spaces, act_space, 128, 256)
    assert [b[2].shape[0] for b in ro8.recurrent_generator(torch.zeros(T, 8, 1), 3)] == [2 * T] * 4
    ro7, got = RolloutStorage(T, 7, ob_space.spaces, act_space, 128, 256), []
    with pytest.raises(IndexError):
        for b in ro7.recurrent_generator(torch.zeros(T, 7, 1), 2):
            got.append(b[2].shape[0])
    assert got == [3 * T, 3 * T]
