"""-m gpu: PPO.update through the HIP path -- cn_ppo_loss_fwd/bwd, cn_adam_clip_step, and the reference's own PPO.update golden
(tests/golden/rollout_*.npz: three losses + post-update weights) replayed on the GPU with every heavy operator a HIP kernel."""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.golden_util import GOLDEN  # noqa: E402


def _ref_losses(values, logp, old_logp, adv, vp, ret, clip, clipped):
    """rl/ppo/ppo.py:66-84 in torch ops (fp64 CPU graph)."""
    ratio = torch.exp(logp - old_logp)
    surr1 = ratio * adv
    surr2 = torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv
    action_loss = -torch.min(surr1, surr2).mean()
    if clipped:
        vpc = vp + (values - vp).clamp(-clip, clip)
        value_loss = 0.5 * torch.max((values - ret).pow(2), (vpc - ret).pow(2)).mean()
    else:
        value_loss = 0.5 * (ret - values).pow(2).mean()
    return value_loss, action_loss


@pytest.mark.parametrize("n,clipped", [(12, True), (61440, True), (5000, False)])
def test_ppo_loss_kernels_match_torch_autograd(n, clipped):
    from crowdnav_prediction_attngraph_amd import hip
    g = torch.Generator().manual_seed(n)
    values = torch.randn(n, 1, generator=g)
    vp = values + 0.3 * torch.randn(n, 1, generator=g)         # both sides of the value clip range
    ret = values + torch.randn(n, 1, generator=g)
    old = -2.0 + 0.5 * torch.randn(n, 1, generator=g)
    logp = old + 0.25 * torch.randn(n, 1, generator=g)        # ratios on both sides of 1 +- 0.2
    adv = torch.randn(n, 1, generator=g)
    vd, ld = values.double().requires_grad_(), logp.double().requires_grad_()
    vl, al = _ref_losses(vd, ld, old.double(), adv.double(), vp.double(), ret.double(), 0.2, clipped)
    (0.5 * vl + al).backward()
    vg, lg = values.cuda().requires_grad_(), logp.cuda().requires_grad_()
    losses = hip.PPOLoss.apply(vg, lg, old.cuda(), adv.cuda(), vp.cuda(), ret.cuda(), 0.2, clipped)
    (0.5 * losses[0] + losses[1]).backward()
    np.testing.assert_allclose(losses.detach().cpu().numpy(), [float(vl), float(al)], rtol=2e-6)
    np.testing.assert_allclose(vg.grad.cpu().numpy(), vd.grad.float().numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), ld.grad.float().numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("n,max_norm", [(2501255, 0.5), (1003, 0.5), (4096, 100.0)])
def test_adam_clip_step_matches_torch(n, max_norm):
    """Three optimiser steps of clip_grad_norm_ + torch.optim.Adam(lr=4e-5, eps=1e-5) on a flat bucket (CPU torch) vs the fused kernel."""
    from crowdnav_prediction_attngraph_amd import hip
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=4e-5, eps=1e-5)
    p, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    norm = torch.zeros(1, device="cuda")
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g)))
        # clip_grad_norm_ with the norm accumulated in fp64 (torch's fp32 norm over 2.5 M elements is only good to ~4e-5; the
        # kernel accumulates block partials in fp64)
        total = grad.double().norm()
        ref.grad = (grad.double() * min(1.0, max_norm / (float(total) + 1e-6))).float()
        opt.step()
        gd = grad.clone().cuda()
        hip.adam_clip_step(p, gd, m, v, step, 4e-5, (0.9, 0.999), 1e-5, max_norm, norm_out=norm)
        assert float(norm) == pytest.approx(float(total), rel=2e-6)
        np.testing.assert_allclose(gd.cpu().numpy(), ref.grad.numpy(), rtol=2e-6, atol=1e-12)   # grads scaled in place like clip_grad_norm_
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=0, atol=5e-7)
    st = opt.state[ref]
    np.testing.assert_allclose(m.cpu().numpy(), st["exp_avg"].numpy(), rtol=1e-4, atol=1e-6 * float(st["exp_avg"].abs().max()))  # lerp cancels near zero
    np.testing.assert_allclose(v.cpu().numpy(), st["exp_avg_sq"].numpy(), rtol=1e-4, atol=1e-6 * float(st["exp_avg_sq"].abs().max()))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz"))), ids=lambda p: os.path.basename(p)[8:-4])
@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
def test_ppo_update_reference_golden_through_the_hip_path(path, mode):
    """The reference's PPO.update on a fixed rollout (losses + post-update parameters, captured by tests/golden/make_golden_policy.py)
    reproduced with the policy, storage and optimiser state on the GPU: evaluate_actions runs the HIP autograd Functions, the
    losses come from cn_ppo_loss_fwd/bwd and the clip + Adam step from cn_adam_clip_step.  Bar: 1e-5 relative."""
    from crowdnav_prediction_attngraph_amd.ppo import PPO
    from tests.test_host_policy import _fill_rollouts, _load_formula, _policy
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    T, E, nmb = meta["T"], meta["E"], meta["nmb"]
    pol, ob_space, act_space = _policy(meta, E, nmb, T)
    _load_formula(pol, meta)
    pol.base.train_gemm_mode = mode
    ro = _fill_rollouts(z, meta, pol, ob_space, act_space)
    pol.cuda()
    ro.to(torch.device("cuda"))
    ro.compute_returns(torch.from_numpy(z["next_value"]).cuda(), True, 0.99, 0.95, False)       # cn_gae
    np.testing.assert_allclose(ro.returns.cpu().numpy()[:-1], z["returns"][:-1], rtol=1e-5, atol=1e-6)
    agent = PPO(pol, 0.2, meta["ppo_epoch"], nmb, 0.5, 0.0, lr=4e-5, eps=1e-5, max_grad_norm=0.5)
    torch.manual_seed(meta["update_seed"])                     # recurrent_generator draws torch.randperm on the CPU generator
    v_loss, a_loss, ent = agent.update(ro)
    np.testing.assert_allclose([v_loss, a_loss, ent], z["losses"], rtol=1e-5, atol=2e-6)
    sd = {k: v.detach().cpu() for k, v in pol.state_dict().items()}
    for k in ("dist.fc_mean.weight", "base.critic_linear.weight", "base.robot_linear.0.weight"):
        np.testing.assert_allclose(sd[k].numpy(), z["after_" + k], rtol=1e-5, atol=2e-6, err_msg=k)
    for k, t in sd.items():
        a = t.numpy().astype(np.float64)
        # every tensor: 256 individual post-update weights at 1e-5 relative (atol: fp32 resolution of the Adam step lr x 4 = 1.6e-4 on
        # weights that are exactly zero in the reference, e.g. biases initialised to 0)
        flat = t.numpy().reshape(-1)
        idx = np.linspace(0, flat.size - 1, min(flat.size, 256)).astype(np.int64)
        np.testing.assert_allclose(flat[idx], z["smp_" + k], rtol=1e-5, atol=2e-6, err_msg="sampled weights of " + k)
        # ... and the sums over the whole tensor, 1e-5 relative to sum |w| (the plain sum cancels)
        np.testing.assert_allclose(np.abs(a).sum(), z["chk_" + k][1], rtol=1e-5, atol=0, err_msg=k)
        np.testing.assert_allclose(a.sum(), z["chk_" + k][0], rtol=0, atol=1e-5 * float(z["chk_" + k][1]) + 1e-12, err_msg=k)
    # every parameter (view into the flat bucket) starts on a 16-byte boundary: their raw pointers feed float4 loads in the kernels
    assert all(p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0 for p in pol.parameters())
    # the torch optimiser object still owns a faithful state (checkpointing): step count and moments of a touched parameter
    st = agent.optimizer.state[pol.dist.fc_mean.weight]
    assert float(st["step"]) == meta["ppo_epoch"] * nmb and float(st["exp_avg"].abs().sum()) > 0
    # the rollout path sees the updated weights (raw-pointer Adam writes invalidate the cn_policy snapshot)
    obs = {k: ro.obs[k][0] for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
    hxs = {"human_node_rnn": ro.recurrent_hidden_states["human_node_rnn"][0], "human_human_edge_rnn": None}
    v_hip, _, _, _ = pol.act(obs, hxs, ro.masks[0], deterministic=True)
    with torch.no_grad():
        v_t, _, _ = pol.base.forward_sequence(obs, hxs["human_node_rnn"], ro.masks[0], 1, E)
    assert torch.allclose(v_hip, v_t, atol=1e-4)
