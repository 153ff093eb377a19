"""-m gpu: the DEFAULT training path (train_fused_hh + train_fused_rn: cn_hh_block_fwd, the per-layer backward kernels of the human-human
block, cn_rn_seq_fwd / cn_rn_seq_bwd, cn_ppo_loss_*, cn_adam_clip_step) against the CPU torch graph at sizes where the kernels pick their
training variants -- the 128 x 512-tile weight-gradient kernel with many splits, every attention size class, the 16 384-row switches, the
multi-split reductions of cn_rn_seq_bwd.  The small-size tests (test_gpu_train.py, test_gpu_ppo.py) pin the same path to the reference's
own goldens; the CPU graph used here is the one tests/test_host_policy.py pins to the reference (rl/networks/model.py:82-90,
selfAttn_srnn_temp_node.py:360-449, rl/ppo/ppo.py:36-101)."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _synth_batch(T, N, H, D, seed):
    """Observation batch [T*N, ...] with ragged detected counts (every attention size class at H = 20), hidden states, done masks, actions."""
    from tests import policy_util as PU
    obs = PU.synth_obs(T * N, H, D, seed=seed)
    rs = np.random.RandomState(seed + 1)
    # mostly few detected humans (like the simulator: ~6 of 20), with a tail up to H
    det = np.minimum(1 + rs.geometric(0.18, size=T * N), H).astype(np.float32)
    det[0], det[1], det[2], det[3] = 1, 8, 9, H
    obs["detected_human_num"] = det.reshape(T * N, 1)
    for e in range(T * N):                                      # rows beyond the detected count carry the simulator's padding value
        obs["spatial_edges"][e, int(det[e]):] = 15.0
    g = torch.Generator().manual_seed(seed + 2)
    h0 = 0.5 * torch.randn(N, 1, 128, generator=g)
    masks = (torch.rand(T * N, 1, generator=g) > 0.05).float()
    actions = torch.randn(T * N, 2, generator=g)
    return {k: torch.from_numpy(v) for k, v in obs.items()}, h0, masks, actions


@pytest.mark.parametrize("N", [64, 512])
def test_default_training_path_matches_the_cpu_graph_in_fp64_at_training_sizes(N):
    """evaluate_actions + a PPO-shaped loss + EVERY parameter gradient, GPU default path (fused, bf16x3) vs the torch-op graph on the CPU in
    fp64, at T = 30 and N = 64 / 512 envs of 20 humans with ragged detected counts (1 920 / 15 360 samples, ~10 k / ~85 k live rows).
    Bars: values and log-probs 1e-4 absolute (north_star); gradients 5e-4 of the tensor's largest entry (measured worst: 3.8e-4, an entry of
    spatial_linear.0.weight, whose gradient passes through the folded out_proj o spatial_linear product) (bf16x3 products: ~2e-5 per layer,
    accumulated over the chain and over up to 15 360 samples)."""
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    torch.manual_seed(5)
    H, D, T = 20, 2, 30
    ob_space, act_space = make_spaces(H, D)
    pol_c = Policy(ob_space.spaces, act_space, base="selfAttn_merge_srnn", base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=N, num_mini_batch=1, seq_length=T))
    pol_g = copy.deepcopy(pol_c).cuda()
    pol_c = pol_c.double()
    assert pol_g.base.train_fused_hh and pol_g.base.train_fused_rn and pol_g.base.train_gemm_mode == "bf16x3"
    obs, h0, masks, actions = _synth_batch(T, N, H, D, seed=100 + N)
    B = T * N
    wv = torch.linspace(-0.5, 1.5, B).view(-1, 1)

    def run(pol, dev, dt):
        o = {k: v.to(dev, dt) for k, v in obs.items()}
        v, lp, ent, hx = pol.evaluate_actions(o, {"human_node_rnn": h0.to(dev, dt)}, masks.to(dev, dt), actions.to(dev, dt))
        loss = (v * wv.to(dev, dt)).mean() + 0.3 * lp.mean() + ent       # mean-shaped, like ppo.py:66-84
        pol.zero_grad()
        loss.backward()
        return (v.detach().cpu().double(), lp.detach().cpu().double(), hx["human_node_rnn"].detach().cpu().double().view(N, -1),
                {k: p.grad.detach().cpu().double() for k, p in pol.named_parameters() if p.grad is not None})

    v_c, lp_c, h_c, g_c = run(pol_c, "cpu", torch.float64)
    v_g, lp_g, h_g, g_g = run(pol_g, "cuda", torch.float32)
    assert float((v_c - v_g).abs().max()) <= 1e-4, float((v_c - v_g).abs().max())
    assert float((lp_c - lp_g).abs().max()) <= 1e-4, float((lp_c - lp_g).abs().max())
    assert float((h_c - h_g).abs().max()) <= 1e-4
    assert set(g_c) == set(g_g)
    worst = ("", 0.0)
    for k in g_c:
        scale = max(float(g_c[k].abs().max()), 1e-6)
        err = float((g_c[k] - g_g[k]).abs().max())
        if err / scale > worst[1]:
            worst = (k, err / scale)
        assert err <= 5e-4 * scale + 1e-7, (k, err, scale)
    # the run is deterministic: a second pass over the same batch gives bit-identical gradients
    _, _, _, g_g2 = run(pol_g, "cuda", torch.float32)
    for k in g_g:
        assert torch.equal(g_g[k], g_g2[k]), k
    print("N=%d worst relative gradient error %.2e (%s)" % (N, worst[1], worst[0]))


def _filled_rollouts(pol, T, E, H, D, seed):
    """A RolloutStorage as a rollout would leave it, filled from synthetic observations: old values / log-probs of the policy itself (plus noise,
    so that ratios and value clips land on both sides of their thresholds), random rewards and episode ends."""
    from crowdnav_prediction_attngraph_amd.policy import make_spaces
    from crowdnav_prediction_attngraph_amd.storage import RolloutStorage
    ob_space, act_space = make_spaces(H, D)
    ro = RolloutStorage(T, E, ob_space.spaces, act_space, 128, 256)
    obs, _, _, _ = _synth_batch(T + 1, E, H, D, seed)
    g = torch.Generator().manual_seed(seed + 9)
    for k in ro.obs:
        if k in obs:
            ro.obs[k].copy_(obs[k].view(T + 1, E, *obs[k].shape[1:]).to(ro.obs[k].dtype))
    ro.recurrent_hidden_states["human_node_rnn"].copy_(0.5 * torch.randn(T + 1, E, 1, 128, generator=g))
    ro.masks.copy_((torch.rand(T + 1, E, 1, generator=g) > 0.04).float())
    ro.actions.copy_(torch.randn(T, E, 2, generator=g))
    ro.rewards.copy_(0.2 * torch.randn(T, E, 1, generator=g))
    with torch.no_grad():       # old statistics from the (CPU) policy, one recurrent pass over each env's trajectory
        flat = {k: ro.obs[k][:T].reshape(T * E, *ro.obs[k].shape[2:]) for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
        v, lp, _, _ = pol.evaluate_actions(flat, {"human_node_rnn": ro.recurrent_hidden_states["human_node_rnn"][0]}, ro.masks[:T].reshape(T * E, 1),
                                           ro.actions.reshape(T * E, 2))
    ro.value_preds[:T].copy_((v + 0.15 * torch.randn(T * E, 1, generator=g)).view(T, E, 1))
    ro.action_log_probs.copy_((lp + 0.2 * torch.randn(T * E, 1, generator=g)).view(T, E, 1))
    next_value = torch.randn(E, 1, generator=g)
    return ro, next_value


def test_ppo_update_at_2x30x256_matches_the_cpu_update():
    """One PPO.update (2 epochs x 2 recurrent minibatches of 30 x 256 samples = four optimiser steps; clipped value loss, grad-norm clip,
    Adam) on the GPU default path vs the same update on the CPU torch path from the same rollout and the same minibatch permutation:
    the three losses at 2e-4 relative (1e-5 absolute) and every post-update weight within 1.6e-5 (mean difference per tensor <= 2e-7; measured: 3 of 16 384 entries of one tensor
    above 2e-6, the largest 5.5e-6)."""
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    from crowdnav_prediction_attngraph_amd.ppo import PPO
    torch.manual_seed(11)
    H, D, T, E, nmb = 20, 2, 30, 512, 2
    ob_space, act_space = make_spaces(H, D)
    pol_c = Policy(ob_space.spaces, act_space, base="selfAttn_merge_srnn", base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=E, num_mini_batch=nmb, seq_length=T))
    pol_c.base.nenv = E
    ro_c, next_value = _filled_rollouts(pol_c, T, E, H, D, seed=7)
    pol_g = copy.deepcopy(pol_c).cuda()
    ro_g = copy.deepcopy(ro_c)
    ro_g.to(torch.device("cuda"))
    out = {}
    for name, pol, ro in (("cpu", pol_c, ro_c), ("gpu", pol_g, ro_g)):
        ro.compute_returns(next_value.to(ro.rewards.device), True, 0.99, 0.95, False)
        agent = PPO(pol, 0.2, 2, nmb, 0.5, 0.0, lr=4e-5, eps=1e-5, max_grad_norm=0.5)
        torch.manual_seed(123)                      # recurrent_generator's randperm (CPU generator on both sides)
        out[name] = (agent.update(ro), {k: v.detach().cpu().double() for k, v in pol.state_dict().items()}, ro.returns.detach().cpu())
    np.testing.assert_allclose(out["gpu"][2].numpy(), out["cpu"][2].numpy(), rtol=1e-5, atol=1e-6)     # cn_gae vs the torch scan
    np.testing.assert_allclose(out["gpu"][0], out["cpu"][0], rtol=2e-4, atol=1e-5)   # (the action loss is a mean of normalised advantages: ~1e-3)
    # Adam divides every gradient entry by its own magnitude: an entry whose gradient is of the size of the two paths' absolute difference
    # (~1e-6 of a tensor whose largest entry is 1e-2) moves by a visibly different step, so the bar per entry is a tenth of the largest
    # possible movement (4 steps x lr = 1.6e-4: a flipped step direction would show as 8e-5), and the MEAN difference per tensor is held at 2e-7
    for k, wc in out["cpu"][1].items():
        d = (out["gpu"][1][k] - wc).abs()
        assert float(d.max()) <= 1.6e-5, (k, float(d.max()))
        assert float(d.mean()) <= 2e-7, (k, float(d.mean()))


@pytest.mark.parametrize("E,T,updates", [(8, 5, 12), (64, 8, 8), (1024, 30, 3)])
def test_training_reruns_are_bit_identical(E, T, updates):
    """Rollout + GAE + PPO.update, `updates` times, twice from the same seed: every weight bit-identical at the end.  The sizes cover
    cn_rn_seq_bwd's single-split reductions (T x N <= 64 samples per minibatch: the class of round 4's scratch race), a mid size, and the
    multi-split / 128 x 512-tile kernels of training batches.  (tests/soak_gpu.py loops 50 updates of the same by hand.)"""
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.trainer import train
    runs = []
    for _ in range(2):
        hist, pol = train("CrowdSimVarNum-v0", num_processes=E, num_steps=T, num_updates=updates, seed=31, config=C.non_randomized(), log=None)
        assert all(np.isfinite([r["value_loss"], r["action_loss"]]).all() for r in hist)
        runs.append(({k: v.detach().clone() for k, v in pol.state_dict().items()}, [(r["value_loss"], r["action_loss"]) for r in hist]))
    assert runs[0][1] == runs[1][1]
    for k, v in runs[0][0].items():
        assert torch.equal(v, runs[1][0][k]), k


@pytest.mark.parametrize("M,K,N,ta,tb", [(512, 512, 512, False, False), (256, 64, 256, True, False), (512, 256, 128, False, False), (512, 512, 1, False, False),
                                          (256, 64, 1, True, False), (33, 70, 45, False, True)])
def test_small_mm_forward_and_both_gradients_match_fp64(M, K, N, ta, tb):
    """cn_small_mm behind hip.weight_mm (the affine folds of the update and their backward: strided operands, slices of a larger matrix,
    vector right-hand sides) against fp64: 1e-6 of the largest entry (exact fp32 products, fixed summation order)."""
    from crowdnav_prediction_attngraph_amd import hip
    g = torch.Generator().manual_seed(M + 3 * K + 7 * N)
    big = torch.randn(3 * (K if ta else M), (M if ta else K) + 5, generator=g)
    a0 = big[(K if ta else M):2 * (K if ta else M), 2:2 + (M if ta else K)]      # a slice: non-contiguous rows
    b0 = torch.randn(*((N, K) if tb else (K, N)), generator=g)
    vec = N == 1 and not tb
    if vec:
        b0 = b0[:, 0]
    dc = torch.randn(M, generator=g) if vec else torch.randn(M, N, generator=g)

    def run(dev, dt):
        a_ = a0.to(dev, dt).requires_grad_()
        b_ = b0.to(dev, dt).requires_grad_()
        aa = a_.t() if ta else a_
        bb = b_.t() if tb else b_
        c = hip.weight_mm(aa, bb) if dev == "cuda" else aa @ bb
        c.backward(dc.to(dev, dt))
        return c.detach().cpu().double(), a_.grad.cpu().double(), b_.grad.cpu().double()

    got, ref = run("cuda", torch.float32), run("cpu", torch.float64)
    for x, y, what in zip(got, ref, ("c", "da", "db")):
        assert x.shape == y.shape, what
        assert float((x - y).abs().max()) <= 2e-6 * max(float(y.abs().max()), 1.0), (what, float((x - y).abs().max()))
    c2 = run("cuda", torch.float32)
    assert all(torch.equal(p, q) for p, q in zip(got, c2))
