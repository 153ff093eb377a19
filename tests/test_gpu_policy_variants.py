"""-m gpu: use_self_attn = False / sort_humans = False through the HIP path (cn_policy_set_self_attention, cn_obs_compact_visible, the
training Functions) on the reference-generated vectors of tests/golden/polvar_*.npz -- see tests/test_policy_variants.py."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests import policy_util as PU  # noqa: E402
from tests.test_policy_variants import PATHS, build, inputs  # noqa: E402


@pytest.mark.parametrize("mode", ["fused", "bf16x3", "fp32"])
@pytest.mark.parametrize("path", PATHS, ids=lambda p: os.path.basename(p)[7:-4])
def test_variant_rollout_forward_matches_the_reference(path, mode):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    N = meta["N"]
    pol = build(meta).cuda()
    pol.load_state_dict({k: torch.from_numpy(v) for k, v in PU.formula_state_dict({k: tuple(v) for k, v in meta["shapes"].items()}).items()})
    pol.rollout_gemm_mode = mode
    obs, hxs = inputs(z, meta, "cuda", n=N)
    masks = torch.from_numpy(z["masks"][:N]).cuda()
    value, action, logp, hx = pol.act(obs, hxs, masks, deterministic=True)
    tol = 1e-4 if mode != "fp32" else 3e-5
    for got, want, what in ((value, z["value"], "value"), (action, z["action"], "action"), (logp, z["logp"], "logp"), (hx["human_node_rnn"], z["hx_out"], "hx")):
        np.testing.assert_allclose(got.cpu().numpy().reshape(want.shape), want, atol=tol, err_msg=what)
    np.testing.assert_allclose(pol.get_value(obs, hxs, masks).cpu().numpy(), z["value"], atol=tol)
    # the taps of the separate-launch path: spatial_linear's output and the robot-human attention on the reference's own row order
    if mode != "fused":
        hip = pol._hip_policy(N, torch.device("cuda", torch.cuda.current_device()))
        hip.set_taps(True)
        pol.act(obs, hxs, masks, deterministic=True)
        taps = hip.taps(N)
        c = pol.base.counted_inputs(obs)
        det = c["detected_human_num"].reshape(N).long().cpu().numpy()
        vm = z["obs_visible_masks"][:N].copy()
        if not meta["sort_humans"]:
            vm[~vm.any(1), 0] = True
        for e in range(N):
            # rows of the device = the visible humans first (index order); the reference keeps them in place
            idx = np.flatnonzero(vm[e]) if not meta["sort_humans"] else np.arange(det[e])
            np.testing.assert_allclose(taps["spatial_lin"][e, :det[e]].cpu().numpy(), z["spatial_lin"][e][idx], atol=tol, err_msg="spatial_lin env %d" % e)
            np.testing.assert_allclose(taps["hr_attn"][e, :det[e]].cpu().numpy(), z["hr_attn"][e][idx], atol=tol, err_msg="hr_attn env %d" % e)
        np.testing.assert_allclose(taps["hr_out"].cpu().numpy(), z["hr_out"], atol=tol)


@pytest.mark.parametrize("path", PATHS, ids=lambda p: os.path.basename(p)[7:-4])
def test_variant_training_forward_and_gradients(path):
    """evaluate_actions on the GPU (default fused robot-node sequence; the human-human block of the variant) vs the reference's outputs, and
    every gradient vs the CPU torch graph of the same module."""
    import copy
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    N = meta["N"]
    pol_c = build(meta)
    pol_c.load_state_dict({k: torch.from_numpy(v) for k, v in PU.formula_state_dict({k: tuple(v) for k, v in meta["shapes"].items()}).items()})
    pol_g = copy.deepcopy(pol_c).cuda()
    res = {}
    for name, pol, dev in (("cpu", pol_c, "cpu"), ("gpu", pol_g, "cuda")):
        obs, hxs = inputs(z, meta, dev)
        v, lp, ent, hx = pol.evaluate_actions(obs, hxs, torch.from_numpy(z["masks"]).to(dev), torch.from_numpy(z["actions"]).to(dev))
        pol.zero_grad()
        (v.mean() + 0.3 * lp.mean()).backward()
        res[name] = (v.detach().cpu().numpy(), lp.detach().cpu().numpy(), hx["human_node_rnn"].detach().cpu().numpy().reshape(N, -1),
                     {k: p.grad.detach().cpu().double() for k, p in pol.named_parameters() if p.grad is not None})
    np.testing.assert_allclose(res["gpu"][0], z["ev_value"], atol=1e-4)
    np.testing.assert_allclose(res["gpu"][1], z["ev_logp"], atol=1e-4)
    np.testing.assert_allclose(res["gpu"][2], z["ev_hx"].reshape(N, -1), atol=1e-4)
    assert set(res["cpu"][3]) == set(res["gpu"][3])
    for k, gc in res["cpu"][3].items():
        scale = max(float(gc.abs().max()), 1e-6)
        err = float((gc - res["gpu"][3][k]).abs().max())
        assert err <= 3e-4 * scale + 1e-7, (k, err, scale)


def test_compact_visible_kernel_matches_the_torch_form():
    from crowdnav_prediction_attngraph_amd.hip import compact_visible
    g = torch.Generator().manual_seed(3)
    for B, H, D in ((1, 1, 2), (37, 20, 2), (513, 64, 12), (9, 33, 12)):
        se = torch.randn(B, H, D, generator=g)
        vm = torch.rand(B, H, generator=g) > 0.6
        vm[0] = False
        if B > 2:
            vm[1] = True
        out, det = compact_visible(se.cuda(), vm.cuda())
        m = vm.clone()
        m[~m.any(1), 0] = True
        order = torch.argsort((~m).to(torch.int8), dim=1, stable=True)
        want = torch.gather(se, 1, order.unsqueeze(-1).expand(-1, -1, D))
        assert torch.equal(out.cpu(), want) and torch.equal(det.cpu().reshape(-1), m.sum(1).float())


@pytest.mark.parametrize("use_self_attn,sort_humans", [(False, True), (True, False), (False, False)])
def test_variants_train_end_to_end_on_the_device(use_self_attn, sort_humans):
    """Fused rollout (the simulator's UNSORTED observation + visible_masks through cn_obs_compact_visible when sort_humans = False) + GAE +
    PPO.update for two updates through trainer.train: the switches reach the policy and the losses are finite."""
    from types import SimpleNamespace
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.trainer import train
    cfg = C.non_randomized(**{"sim.human_num": 8})
    cfg.args = SimpleNamespace(**dict(vars(cfg.args), sort_humans=sort_humans))
    hist, pol = train("CrowdSimVarNum-v0", num_processes=32, num_steps=10, num_updates=2, seed=9, config=cfg, log=None, use_self_attn=use_self_attn)
    assert pol.base.use_self_attn == use_self_attn and pol.base.sort_humans == sort_humans
    for r in hist:
        assert all(np.isfinite([r["value_loss"], r["action_loss"], r["entropy"]]))


from tests.test_policy_variants import ROLL, check_update_against_the_reference, filled_rollouts  # noqa: E402


@pytest.mark.parametrize("path", ROLL, ids=lambda p: os.path.basename(p)[8:-4])
def test_variant_ppo_update_reference_golden_through_the_hip_path(path):
    """The reference's PPO.update with the switches set, replayed on the GPU (cn_obs_compact_visible / the spatial MLP in front of the fused
    robot-node sequence, HIP losses, clip + Adam): 1e-5 relative like the default configuration's golden (tests/test_gpu_ppo.py)."""
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    ob_space, act_space = make_spaces(meta["H"], meta["D"])
    pol = Policy(ob_space.spaces, act_space, base="selfAttn_merge_srnn",
                 base_kwargs=dict(env_name=meta["env_name"], num_processes=meta["E"], num_mini_batch=meta["nmb"], seq_length=meta["T"],
                                  use_self_attn=meta["use_self_attn"], sort_humans=meta["sort_humans"]))
    pol.load_state_dict({k: torch.from_numpy(v) for k, v in PU.formula_state_dict({k: tuple(v) for k, v in meta["shapes"].items()}).items()})
    pol.cuda()
    check_update_against_the_reference(z, meta, pol, filled_rollouts(z, meta, "cuda"), atol=2e-6, rtol=1e-5)
