"""-m gpu: the host mirror end to end on the device -- reference-compatible vec-env API, fused rollout, PPO update."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_vec_env_api_contract():
    from crowdnav_prediction_attngraph_amd import config as C, info as I
    from crowdnav_prediction_attngraph_amd.vec_env import make_vec_envs
    envs = make_vec_envs("CrowdSimVarNum-v0", 425, 16, 0.99, None, torch.device("cuda"), False, config=C.non_randomized())
    assert envs.action_space.shape[0] == 2 and envs.action_space.__class__.__name__ == "Box"
    assert envs.observation_space.spaces["spatial_edges"].shape == (20, 2)
    obs = envs.reset()
    assert obs["robot_node"].shape == (16, 1, 7) and obs["visible_masks"].dtype == torch.bool and obs["spatial_edges"].is_cuda
    n_ep = 0
    for t in range(230):
        obs, reward, done, infos = envs.step(torch.full((16, 2), 0.05, device="cuda"))
        assert reward.shape == (16, 1) and not reward.is_cuda and reward.dtype == torch.float32
        assert isinstance(done, np.ndarray) and done.dtype == bool and len(infos) == 16
        for d, inf in zip(done, infos):
            assert ("episode" in inf) == bool(d)
            if d:
                n_ep += 1
                assert isinstance(inf["info"], (I.Timeout, I.Collision, I.ReachGoal)) and inf["episode"]["l"] >= 1
    assert n_ep >= 16        # everything times out after 197 steps at the latest
    assert envs.talk2Env(None) == [True] * 16
    envs.close()
    with pytest.raises(NotImplementedError):   # settings the reference itself cannot run fail loudly instead of falling back (tests/test_host_config.py)
        make_vec_envs("CrowdSimPred-v0", 425, 16, 0.99, None, torch.device("cuda"), False,
                      config=C.Config(**{"sim.predict_method": "const_vel", "robot.visible": True}))
    with pytest.raises(NotImplementedError):
        make_vec_envs("CrowdSimVarNum-v0", 425, 16, 0.99, None, torch.device("cuda"), False, config=C.Config(**{"sim.human_num": 60, "sim.human_num_range": 10}))
    # a varying crowd (sim.human_num_range) and the unicycle robot run on the device: observations carry human_num + range rows
    var = make_vec_envs("CrowdSimVarNum-v0", 425, 8, 0.99, None, torch.device("cuda"), False,
                        config=C.Config(**{"sim.human_num": 6, "sim.human_num_range": 5, "action_space.kinematics": "unicycle"}))
    ob = var.reset()
    assert ob["spatial_edges"].shape == (8, 11, 2) and var.observation_space.spaces["spatial_edges"].shape == (11, 2)
    for t in range(30):
        ob, rew, done, infos = var.step(torch.full((8, 2), 0.03, device="cuda"))
    assert float(ob["detected_human_num"].min()) >= 1 and float(ob["detected_human_num"].max()) <= 11
    var.close()
    one = make_vec_envs("CrowdSimVarNum-v0", 425, 1, 0.99, None, torch.device("cuda"), False)   # num_processes=1 -> phase 'test' (envs.py:55-58)
    assert one.cfg.phase == 2
    ob = one.reset()
    ob, rew, done, infos = one.step(torch.zeros(1, 2))
    assert ob["robot_node"].shape == (1, 1, 7) and isinstance(infos[0]["info"], (I.Nothing, I.Danger))
    one.close()


def test_policy_module_uses_hip_for_act_and_tracks_weight_updates():
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    from tests import policy_util as PU
    torch.manual_seed(0)
    ob_space, act_space = make_spaces(20, 2)
    pol = Policy(ob_space.spaces, act_space, base="selfAttn_merge_srnn", base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=32)).cuda()
    obs = {k: torch.from_numpy(v).cuda() for k, v in PU.synth_obs(32, 20, 2, 1).items()}
    hxs = {"human_node_rnn": torch.randn(32, 1, 128, device="cuda"), "human_human_edge_rnn": torch.zeros(32, 21, 256, device="cuda")}
    masks = torch.ones(32, 1, device="cuda")
    v1, a1, lp1, h1 = pol.act(obs, hxs, masks, deterministic=True)
    with torch.no_grad():  # torch path (training graph) on the same inputs
        v_t, feat, h_t = pol.base.forward_sequence(obs, hxs["human_node_rnn"], masks, 1, 32)
        mean = pol.dist.fc_mean(feat)
    assert torch.allclose(v1, v_t, atol=1e-4) and torch.allclose(a1, mean, atol=1e-4) and torch.allclose(h1["human_node_rnn"].view(32, 128), h_t, atol=1e-4)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.01 * torch.randn_like(p))
    v2, _, _, _ = pol.act(obs, hxs, masks, deterministic=True)
    with torch.no_grad():
        v_t2, _, _ = pol.base.forward_sequence(obs, hxs["human_node_rnn"], masks, 1, 32)
    assert not torch.allclose(v1, v2, atol=1e-5) and torch.allclose(v2, v_t2, atol=1e-4)   # the HIP snapshot was refreshed


def test_fused_rollout_and_update_run_and_learn_signal_is_finite():
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.trainer import train
    hist, pol = train("CrowdSimVarNum-v0", num_processes=64, num_steps=30, num_updates=3, config=C.non_randomized(**{"sim.human_num": 5}), log=None)
    assert len(hist) == 3
    for r in hist:
        assert all(np.isfinite([r["value_loss"], r["action_loss"], r["entropy"]])) and r["samples_per_s"] > 0
    assert abs(hist[0]["entropy"] - 1.4189385) < 1e-3          # 0.5 + 0.5 log(2 pi) with logstd = 0
    assert sum(r["episodes"] for r in hist) > 0
    hist2, _ = train("CrowdSimPred-v0", num_processes=32, num_steps=8, num_updates=1, config=C.Config(**{"sim.human_num": 6, "sim.predict_method": "const_vel"}), log=None)
    assert np.isfinite(hist2[0]["value_loss"])


def test_fused_rollout_equals_reference_style_stepping():
    """collect_rollout (zero-copy, no sync) fills the storage exactly like act()/envs.step()/insert() does."""
    from crowdnav_prediction_attngraph_amd import config as C
    from crowdnav_prediction_attngraph_amd.policy import Policy
    from crowdnav_prediction_attngraph_amd.storage import RolloutStorage
    from crowdnav_prediction_attngraph_amd.trainer import collect_rollout
    from crowdnav_prediction_attngraph_amd.vec_env import make_vec_envs
    E, T = 24, 12
    cfg = C.non_randomized(**{"sim.human_num": 8})
    torch.manual_seed(1)
    ea = make_vec_envs("CrowdSimVarNum-v0", 9, E, 0.99, None, torch.device("cuda"), False, config=cfg)
    eb = make_vec_envs("CrowdSimVarNum-v0", 9, E, 0.99, None, torch.device("cuda"), False, config=cfg)
    pol = Policy(ea.observation_space.spaces, ea.action_space, base="selfAttn_merge_srnn", base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=E)).cuda()
    ra = RolloutStorage(T, E, ea.observation_space.spaces, ea.action_space, 128, 256); ra.to("cuda")
    rb = RolloutStorage(T, E, eb.observation_space.spaces, eb.action_space, 128, 256); rb.to("cuda")
    oa, ob = ea.reset(), eb.reset()
    for k in ra.obs:
        ra.obs[k][0].copy_(oa[k]); rb.obs[k][0].copy_(ob[k])
    g = torch.Generator(device="cuda").manual_seed(3)
    collect_rollout(ea, pol, ra, generator=g)
    torch.manual_seed(0)
    g = torch.Generator(device="cuda").manual_seed(3)
    hip_pol = pol._hip_policy(E, torch.device("cuda", torch.cuda.current_device()))
    eps_all = torch.empty(T, E, 2, device="cuda").normal_(generator=g)   # collect_rollout draws the noise of a rollout in one call
    for t in range(T):
        obs_t = {k: rb.obs[k][t] for k in ("robot_node", "temporal_edges", "spatial_edges", "detected_human_num")}
        eps = eps_all[t]
        # (the row plan the simulator wrote beside this observation: same tile composition, hence the same fp32 summation order, as in
        # collect_rollout -- without it the two rollouts agree to ~1e-6 instead of bit for bit)
        out = hip_pol.act(obs_t, rb.recurrent_hidden_states["human_node_rnn"][t], rb.masks[t], eps=eps, row_plan=eb._env.row_plan)
        obs, reward, done, infos = eb.step(out["action"])
        masks = torch.FloatTensor([[0.0] if d else [1.0] for d in done])
        rb.insert(obs, {"human_node_rnn": out["hxs"]}, out["action"], out["logp"], out["value"], reward, masks, torch.ones(E, 1))
    for name in ("rewards", "value_preds", "actions", "action_log_probs", "masks"):
        assert torch.equal(getattr(ra, name), getattr(rb, name)), name
    for k in ra.obs:
        assert torch.equal(ra.obs[k], rb.obs[k]), k
    assert torch.equal(ra.recurrent_hidden_states["human_node_rnn"], rb.recurrent_hidden_states["human_node_rnn"])


@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
def test_hip_attention_forward_backward_matches_torch_autograd(mode):
    """evaluate_actions on the GPU (HH attention core = cn_hh_attention_fwd/bwd; the three large Linear layers =
    cn_linear_fwd / cn_linear_wgrad in 'bf16x3' mode, rocBLAS fp32 in 'fp32' mode) vs the pure torch-op graph on CPU:
    same values, log-probs and parameter gradients."""
    import copy
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    from tests import policy_util as PU
    torch.manual_seed(3)
    H, D, T, N = 20, 2, 5, 6
    ob_space, act_space = make_spaces(H, D)
    pol_c = Policy(ob_space.spaces, act_space, base="selfAttn_merge_srnn", base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=N, num_mini_batch=1, seq_length=T))
    pol_g = copy.deepcopy(pol_c).cuda()
    pol_g.base.train_gemm_mode = mode
    obs = PU.synth_obs(T * N, H, D, seed=4)
    obs_c = {k: torch.from_numpy(v) for k, v in obs.items()}
    h0 = torch.randn(N, 1, 128)
    masks = (torch.rand(T * N, 1) > 0.2).float()
    actions = torch.randn(T * N, 2)

    def run(pol, dev):
        o = {k: v.to(dev) for k, v in obs_c.items()}
        v, lp, ent, _ = pol.evaluate_actions(o, {"human_node_rnn": h0.to(dev)}, masks.to(dev), actions.to(dev))
        loss = (v * torch.linspace(-1, 1, T * N, device=dev).view(-1, 1)).sum() + (lp * 0.3).sum() + ent
        pol.zero_grad()
        loss.backward()
        return v.detach().cpu(), lp.detach().cpu(), {k: p.grad.detach().cpu() for k, p in pol.named_parameters() if p.grad is not None}

    v_c, lp_c, g_c = run(pol_c, "cpu")
    v_g, lp_g, g_g = run(pol_g, "cuda")
    assert torch.allclose(v_c, v_g, atol=1e-4) and torch.allclose(lp_c, lp_g, atol=1e-4)
    assert set(g_c) == set(g_g)
    for k in g_c:
        scale = max(float(g_c[k].abs().max()), 1e-3)
        assert float((g_c[k] - g_g[k]).abs().max()) <= 2e-4 * scale + 1e-5, (k, float((g_c[k] - g_g[k]).abs().max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("H,B", [(20, 96), (64, 48), (5, 120)])   # the fp64 reference is a Python loop over samples: keep B small
def test_hh_attention_size_classes_match_fp64_autograd(H, B):
    """cn_hh_attention_fwd/bwd through their size-class lists (<= 8 / 16 / 32 / 64 live humans, one launch per class walking only
    its own units) vs an fp64 torch graph of softmax(scale q k^T) v per (sample, head): every class is populated."""
    from crowdnav_prediction_attngraph_amd import hip
    g = torch.Generator().manual_seed(H * 1000 + B)
    nd = torch.randint(1, H + 1, (B,), generator=g)
    nd[:4] = torch.tensor([1, min(H, 8), min(H, 9), H])
    row_off = torch.zeros(B + 1, dtype=torch.int32)
    row_off[1:] = torch.cumsum(nd, 0)
    R = int(row_off[-1])
    qkv = torch.randn(R, 1536, generator=g)
    d_out = torch.randn(R, 512, generator=g)
    qg = qkv.cuda().requires_grad_()
    out = hip.HHAttention.apply(qg, row_off.cuda(), B, H, 0.125)
    out.backward(d_out.cuda())
    torch.cuda.synchronize()
    qr = qkv.double().requires_grad_()
    outs = []
    for b in range(B):
        r0, r1 = int(row_off[b]), int(row_off[b + 1])
        blk = qr[r0:r1].view(r1 - r0, 3, 8, 64)
        q, k, v = blk[:, 0].transpose(0, 1), blk[:, 1].transpose(0, 1), blk[:, 2].transpose(0, 1)     # [8, nd, 64]
        p = torch.softmax(0.125 * q @ k.transpose(1, 2), dim=-1)
        outs.append((p @ v).transpose(0, 1).reshape(r1 - r0, 512))
    ref = torch.cat(outs)
    ref.backward(d_out.double())
    assert float((out.detach().cpu() - ref.detach().float()).abs().max()) <= 2e-5
    assert float((qg.grad.cpu() - qr.grad.float()).abs().max()) <= 5e-5 * max(1.0, float(qr.grad.abs().max()))
    # the lists only steer which launch handles a unit: a second run (lists rebuilt in another atomic order) is bit-identical
    qg2 = qkv.cuda().requires_grad_()
    out2 = hip.HHAttention.apply(qg2, row_off.cuda(), B, H, 0.125)
    out2.backward(d_out.cuda())
    assert torch.equal(out, out2) and torch.equal(qg.grad, qg2.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,relu", [(1000, 512, 128, True), (4097, 1536, 512, False), (300, 256, 512, True), (64, 128, 128, False),
                                         (20000, 512, 128, True),
                                         # >= 32768 rows: the pipelined weight-gradient kernel (whole 32-row tiles; the row tail goes to the
                                         # two-barrier kernel as one more split), 128 x 512, 128 x 256 and 128 x 128 tiles
                                         (40001, 256, 512, True), (33000, 384, 128, False), (36000, 256, 256, True), (34017, 128, 512, False)])
def test_hip_linear_forward_and_gradients_match_fp64(M, N, K, relu):
    """cn_linear_fwd / cn_linear_wgrad (bf16x3 split precision) against an fp64 torch graph: outputs and all three
    gradients within 1e-4 of the largest reference magnitude (the fp32 reference itself sits at ~1e-6)."""
    from crowdnav_prediction_attngraph_amd import hip
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    dy = torch.randn(M, N, generator=g)
    xg, wg, bg = (t.cuda().requires_grad_() for t in (x, w, b))
    assert hip.linear_supported(xg, wg)
    y = hip.HipLinear.apply(xg, wg, bg, relu)
    y.backward(dy.cuda())
    torch.cuda.synchronize()
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    yr = torch.nn.functional.linear(xr, wr, br)
    if relu:
        # ReLU is discontinuous in its gradient: outputs within rounding of zero may land on either side, so the reference
        # uses the kernel's own active set (and checks it only disagrees with the fp64 one where |y| is at rounding level)
        act = (y.detach().cpu() > 0)
        flipped = yr.detach()[act != (yr.detach() > 0)].abs()
        assert flipped.numel() == 0 or float(flipped.max()) < 1e-4
        yr = yr * act.double()
    yr.backward(dy.double())

    def close(a, ref, what):
        ref = ref.float()
        err = float((a.cpu() - ref).abs().max())
        assert err <= 1e-4 * max(float(ref.abs().max()), 1e-3), (what, err, float(ref.abs().max()))

    close(y.detach(), yr.detach(), "y")
    close(xg.grad, xr.grad, "dx")
    close(wg.grad, wr.grad, "dw")
    close(bg.grad, br.grad, "db")
    # deterministic: a second backward gives bit-identical gradients
    xg2, wg2, bg2 = (t.cuda().requires_grad_() for t in (x, w, b))
    hip.HipLinear.apply(xg2, wg2, bg2, relu).backward(dy.cuda())
    assert torch.equal(wg.grad, wg2.grad) and torch.equal(bg.grad, bg2.grad) and torch.equal(xg.grad, xg2.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,act,relu_from,pad_to", [(61440, 256, 256, 2, None, 0), (5000, 512, 128, 2, None, 0), (4099, 320, 256, 0, 256, 384),
                                                          (40000, 256, 320, 3, None, 0), (777, 128, 384, 3, None, 0), (33001, 256, 256, 4, None, 0),
                                                          (2048, 128, 512, 0, None, 0), (9000, 256, 64, 1, None, 0)])
def test_linear_act_epilogues_match_fp64(M, N, K, act, relu_from, pad_to):
    """cn_linear_fwd_act / cn_split_bf16_padded (the robot-node sequence's products on the bf16x3 kernel: tanh, times relu'(aux), times
    tanh'(aux), a ReLU column range, a weight padded to whole 128-column tiles; A and aux with row strides wider than the product) against
    fp64, at the shapes cn_rn_seq_fwd / cn_rn_seq_bwd use: within 1e-4 of the largest reference magnitude (the bar of the big layers' test above;
    measured: 2.3e-5 on 15 M tanh outputs whose arguments reach +-5)."""
    from crowdnav_prediction_attngraph_amd import hip
    g = torch.Generator().manual_seed(M + 7 * N + K + act)
    xw = torch.randn(M, K + 64, generator=g)                       # the product reads a column slice of a wider buffer
    x = xw[:, 32:32 + K]
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1 if act < 3 else None
    auxw = torch.tanh(torch.randn(M, N + 128, generator=g)) * (torch.rand(M, N + 128, generator=g) > 0.3)
    aux = auxw[:, 64:64 + N]
    y = hip.linear_act(xw.cuda()[:, 32:32 + K], w.cuda(), b.cuda() if b is not None else None, act, aux=auxw.cuda()[:, 64:64 + N] if act >= 3 else None,
                       relu_from=relu_from, pad_to=pad_to)
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t()
    if b is not None:
        ref = ref + b.double()
    if act == 1:
        ref = ref.clamp(min=0)
    elif act == 2:
        ref = torch.tanh(ref)
    elif act == 3:
        ref = ref * (aux > 0).double()
    elif act == 4:
        ref = ref * (1.0 - aux.double() ** 2)
    if relu_from is not None:
        ref[:, relu_from:] = ref[:, relu_from:].clamp(min=0)
    got = y.cpu().double()
    assert got.shape == (M, pad_to or N)
    err = float((got[:, :N] - ref).abs().max())
    assert err <= 1e-4 * max(float(ref.abs().max()), 1.0), (err, float(ref.abs().max()))
    if pad_to:
        assert float(got[:, N:].abs().max()) == 0.0               # zero rows of the padded weight, zero bias, ReLU range


@pytest.mark.gpu
def test_hr_attention_forward_backward_matches_dense_torch():
    """cn_hr_attention_fwd/bwd on compacted rows, in the u = Ws^T t form, vs the reference's dense masked formulation
    (att_func, selfAttn_srnn_temp_node.py:145-177: t . (Ws o + bs) * H/8, masked_fill(-1e9), softmax, bmm) under torch
    autograd in fp64: same output and same gradients for t, Ws and o (bs gets zero: the softmax cannot see it)."""
    from crowdnav_prediction_attngraph_amd import hip
    g = torch.Generator().manual_seed(11)
    B, H = 37, 20
    nd = torch.randint(1, H + 1, (B,), generator=g)
    nd[0], nd[1] = 1, H
    row_off = torch.cat([torch.zeros(1, dtype=torch.int64), nd.cumsum(0)]).to(torch.int32)
    R = int(row_off[-1])
    t = torch.randn(B, 64, generator=g)
    Ws = torch.randn(64, 256, generator=g) * 0.05
    bs = torch.randn(64, generator=g)
    o = torch.randn(R, 256, generator=g)
    d_hr = torch.randn(B, 256, generator=g)
    tg, wg, og = (x.cuda().requires_grad_() for x in (t, Ws, o))
    hr = hip.HRAttention.apply(tg @ wg, og, row_off.cuda(), H)
    hr.backward(d_hr.cuda())
    # dense fp64 reference
    tr, wr, br, orr = (x.double().requires_grad_() for x in (t, Ws, bs, o))
    idx = torch.cat([torch.arange(int(n)) + b * H for b, n in enumerate(nd)])
    O = torch.zeros(B * H, 256, dtype=torch.float64).index_copy(0, idx, orr).view(B, H, 256)
    S = torch.nn.functional.linear(O, wr, br)
    valid = torch.arange(H).view(1, H) < nd.view(B, 1)
    a = (tr.unsqueeze(1) * S).sum(-1) * (H / 8.0)
    a = torch.softmax(a.masked_fill(~valid, -1e9), dim=-1)
    ref = torch.bmm(a.unsqueeze(1), O).squeeze(1)
    ref.backward(d_hr.double())
    assert float(br.grad.abs().max()) < 1e-9
    for got, want, what in ((hr.detach(), ref.detach(), "hr"), (tg.grad, tr.grad, "d_t"), (wg.grad, wr.grad, "d_Ws"), (og.grad, orr.grad, "d_o")):
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 2e-5 * max(float(want.abs().max()), 1.0), (what, err)


@pytest.mark.gpu
@pytest.mark.parametrize("T,N", [(5, 6), (30, 70), (7, 32), (3, 129)])
def test_gru_sequence_kernels_match_torch_autograd(T, N):
    """cn_gru_seq_fwd / cn_gru_seq_bwd (one launch per direction, W_hh in registers) against the step-by-step torch graph of
    AttnGraphBase._gru_cell in fp64: states and every gradient (gi, h0, W_hh, b_hh), with done masks in the sequence."""
    from crowdnav_prediction_attngraph_amd import hip
    g = torch.Generator().manual_seed(T * 1000 + N)
    gi = torch.randn(T, N, 384, generator=g)
    h0 = torch.randn(N, 128, generator=g)
    m = (torch.rand(T, N, 1, generator=g) > 0.25).float()
    w = torch.randn(384, 128, generator=g) / 128 ** 0.5
    b = torch.randn(384, generator=g) * 0.1
    d_hs = torch.randn(T, N, 128, generator=g)
    gi_g, h0_g, w_g, b_g = (x.cuda().requires_grad_() for x in (gi, h0, w, b))
    hs = hip.GRUSequence.apply(gi_g, h0_g, m.cuda(), w_g, b_g)
    hs.backward(d_hs.cuda())
    gi_r, h0_r, w_r, b_r = (x.double().requires_grad_() for x in (gi, h0, w, b))
    h, out = h0_r, []
    for t in range(T):
        hm = h * m[t].double()
        gh = torch.nn.functional.linear(hm, w_r, b_r)
        i_r, i_z, i_n = gi_r[t].chunk(3, -1)
        h_r, h_z, h_n = gh.chunk(3, -1)
        r, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1.0 - z) * n + z * hm
        out.append(h)
    ref = torch.stack(out, 0)
    ref.backward(d_hs.double())
    for got, want, what in ((hs.detach(), ref.detach(), "hs"), (gi_g.grad, gi_r.grad, "d_gi"), (h0_g.grad, h0_r.grad, "d_h0"),
                            (w_g.grad, w_r.grad, "d_Whh"), (b_g.grad, b_r.grad, "d_bhh")):
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 2e-5 * max(float(want.abs().max()), 1.0), (what, err, float(want.abs().max()))


@pytest.mark.gpu
def test_skinny_products_of_the_update_match_fp64():
    """The update's replacements for library products that reduce tens of thousands of rows into a tiny matrix: RightMatmul (t @ w with
    the weight gradient on the split-K kernel) and _skinny_linear (one / two output columns as multiply + reduce), values and gradients
    against fp64."""
    from crowdnav_prediction_attngraph_amd import hip, policy
    g = torch.Generator().manual_seed(5)
    M = 5000
    t, w, du = torch.randn(M, 64, generator=g), torch.randn(64, 256, generator=g) * 0.1, torch.randn(M, 256, generator=g)
    tg, wg = t.cuda().requires_grad_(), w.cuda().requires_grad_()
    u = hip.RightMatmul.apply(tg, wg)
    u.backward(du.cuda())
    tr, wr = t.double().requires_grad_(), w.double().requires_grad_()
    (tr @ wr).backward(du.double())

    def close(a, ref, what, tol=1e-4):
        ref = ref.float()
        err = float((a.detach().cpu() - ref).abs().max())
        assert err <= tol * max(float(ref.abs().max()), 1e-3), (what, err, float(ref.abs().max()))

    close(u, (tr @ wr).detach(), "u")
    close(tg.grad, tr.grad, "dt")
    close(wg.grad, wr.grad, "dw")
    for n_out in (1, 2):
        x, w2, b2, dy = torch.randn(M, 256, generator=g), torch.randn(n_out, 256, generator=g) * 0.1, torch.randn(n_out, generator=g), torch.randn(M, n_out, generator=g)
        xg, w2g, b2g = x.cuda().requires_grad_(), w2.cuda().requires_grad_(), b2.cuda().requires_grad_()
        y = policy._skinny_linear(xg, w2g, b2g)
        assert y.shape == (M, n_out)
        y.backward(dy.cuda())
        xr, w2r, b2r = x.double().requires_grad_(), w2.double().requires_grad_(), b2.double().requires_grad_()
        yr = torch.nn.functional.linear(xr, w2r, b2r)
        yr.backward(dy.double())
        close(y, yr.detach(), "y%d" % n_out, 1e-5)
        close(xg.grad, xr.grad, "dx%d" % n_out, 1e-5)
        close(w2g.grad, w2r.grad, "dw%d" % n_out, 1e-5)
        close(b2g.grad, b2r.grad, "db%d" % n_out, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,D", [(3000, 20, 2), (517, 48, 12), (40, 5, 2)])
def test_fused_hh_block_forward_equals_the_per_layer_kernels(B, H, D):
    """cn_hh_block_fwd (hip.HHBlockFused: the rollout's fused kernel on the training weights + the activations the backward needs) against
    the five-launch path (Embed0 -> HipLinear -> HipLinear -> HHAttention -> HipLinear) on ragged crowds: same output rows, same gradient
    of every parameter of the block (the backward runs the same per-layer kernels on the fused forward's saved activations, so any row the
    fused kernel failed to write shows up here), and nothing written outside the live rows (guard rows behind every output buffer)."""
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    torch.manual_seed(3)
    ob_space, act_space = make_spaces(H, D)
    env_name = "CrowdSimVarNum-v0" if D == 2 else "CrowdSimPred-v0"
    net = Policy(ob_space.spaces, act_space, base_kwargs=dict(env_name=env_name, num_processes=16), base="selfAttn_merge_srnn").cuda()
    base = net.base
    g = torch.Generator(device="cuda").manual_seed(B)
    se = torch.randn(B, H, D, device="cuda", generator=g)
    det = torch.clamp((torch.rand(B, device="cuda", generator=g) ** 2 * (H + 1)).long(), 1, H)      # most envs see few humans, some all
    det[::7] = H
    wgt = torch.randn(int(det.sum()), 256, device="cuda", generator=g)
    res = {}
    for fused in (False, True):
        base.train_fused_hh = fused
        for p in net.parameters():
            p.grad = None
        o, ro = base._hh_block(se, det)
        (o * wgt).sum().backward()
        res[fused] = (o.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    assert int(ro[-1]) == int(det.sum()) and res[True][0].shape == res[False][0].shape
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-5
    assert set(res[True][1]) == set(res[False][1]) and len(res[True][1]) >= 14
    for k, gref in res[False][1].items():
        # the ReLU masks (e0 > 0, x > 0, out_sp > 0) come from forward values that the two paths round differently (FMA contraction,
        # summation order): a few of the 10^5..10^6 entries within rounding of zero get the other mask, and ONE flipped entry moves a
        # weight-gradient entry by a whole |dy x| term.  Entry-wise bars are therefore meaningless here; the Frobenius distance is not
        # (unwritten activations are caught exactly further down)
        if k.endswith("k_linear.bias"):
            continue                     # softmax ignores a key bias: its gradient is rounding noise around an exact zero in both paths
        err = float((res[True][1][k] - gref).norm()) / (float(gref.norm()) + 1e-6)
        assert err <= 5e-3, (k, err)
    # raw ABI call with guard rows: every live row written, nothing behind them touched
    L = A.lib()
    nd = det.to(torch.int32)
    row_off = torch.cat([nd.new_zeros(1), nd.cumsum(0, dtype=torch.int32)])
    R, G = int(row_off[-1]), 128
    sa = base.spatial_attn
    ws = torch.empty(int(L.cn_hh_block_workspace_bytes()), dtype=torch.uint8, device="cuda")
    w = [sa.embedding_layer[0].weight, sa.embedding_layer[0].bias, sa.embedding_layer[2].weight, sa.embedding_layer[2].bias,
         torch.randn(1536, 512, device="cuda", generator=g) * 0.05, torch.randn(1536, device="cuda", generator=g),
         torch.randn(256, 512, device="cuda", generator=g) * 0.05, torch.randn(256, device="cuda", generator=g)]
    w = [t.detach().contiguous() for t in w]
    outs = [torch.full((R + G, n), 7777.0, device="cuda") for n in (128, 512, 1536, 512, 256)]
    A.check(L.cn_hh_block_fwd(B, H, D, A.ptr(se), A.ptr(row_off), *[A.ptr(t) for t in w], 0.125, A.ptr(ws), *[A.ptr(t) for t in outs], A.stream_ptr()),
            "cn_hh_block_fwd")
    torch.cuda.synchronize()
    for name, t in zip(("e0", "x", "qkv", "attn", "out_sp"), outs):
        assert int((t[R:] != 7777.0).sum()) == 0, "%s: rows behind the live rows were written" % name
        assert int((t[:R] == 7777.0).sum()) == 0, "%s: live entries left unwritten" % name


def test_episode_stats_kernel_equals_the_torch_expression():
    """trainer.EpisodeStats.update on the step outputs of the simulator (uint8 done / info, float64 returns, int32 lengths): one fixed-order
    launch (cn_episode_stats_update) against the torch-op form it replaces; and identical bits on a repeat (no atomics)."""
    from crowdnav_prediction_attngraph_amd.trainer import EpisodeStats
    g = torch.Generator(device="cuda").manual_seed(4)
    E = 4096
    a, b = EpisodeStats("cuda"), EpisodeStats("cuda")
    ref = torch.zeros(8, dtype=torch.float64, device="cuda")
    for t in range(5):
        done = (torch.rand(E, device="cuda", generator=g) < 0.02).to(torch.uint8)
        info = torch.randint(0, 5, (E,), device="cuda", generator=g).to(torch.uint8)
        ep_ret = torch.randn(E, device="cuda", generator=g, dtype=torch.float64) * 10
        ep_len = torch.randint(1, 200, (E,), device="cuda", generator=g).to(torch.int32)
        a.update(done, info, ep_ret, ep_len)
        b.update(done, info, ep_ret, ep_len)
        d = done.to(torch.float64)
        ref[0] += d.sum(); ref[1] += (ep_ret * d).sum(); ref[2] += (ep_len.to(torch.float64) * d).sum()
        for code in (1, 2, 3):
            ref[2 + code] += ((info == code).to(torch.float64) * d).sum()
    assert torch.equal(a.acc, b.acc)
    assert torch.allclose(a.acc, ref, rtol=1e-12, atol=1e-9), (a.acc, ref)
    assert float(a.acc[0]) > 0


@pytest.mark.parametrize("T,N,H", [(30, 96, 20), (7, 33, 5)])
def test_fused_robot_node_sequence_equals_the_module_path(T, N, H):
    """evaluate_actions with the robot-node sequence as ONE forward and ONE backward call (hip.RnSequence: cn_rn_seq_fwd / cn_rn_seq_bwd;
    rl/networks/model.py:82-90 -> selfAttn_srnn_temp_node.py:395-449) against the same weights run through the torch modules with the
    per-op HIP Functions (train_fused_rn = False): values, log-probs, entropy and the gradient of EVERY parameter."""
    import copy
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    from tests import policy_util as PU
    torch.manual_seed(11)
    D = 2
    ob_space, act_space = make_spaces(H, D)
    a = Policy(ob_space.spaces, act_space, base="selfAttn_merge_srnn", base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=N, num_mini_batch=1, seq_length=T)).cuda()
    with torch.no_grad():
        a.dist.logstd._bias.copy_(torch.tensor([[-0.3], [0.2]]))
    b = copy.deepcopy(a)
    a.base.train_fused_rn, b.base.train_fused_rn = True, False
    obs = {k: torch.from_numpy(v).cuda() for k, v in PU.synth_obs(T * N, H, D, seed=5).items()}
    g = torch.Generator(device="cuda").manual_seed(2)
    h0 = torch.randn(N, 1, 128, device="cuda", generator=g)
    masks = (torch.rand(T * N, 1, device="cuda", generator=g) > 0.15).float()
    actions = torch.randn(T * N, 2, device="cuda", generator=g)
    wv = torch.randn(T * N, 1, device="cuda", generator=g)
    wl = torch.randn(T * N, 1, device="cuda", generator=g)

    def run(pol):
        v, lp, ent, hx = pol.evaluate_actions(obs, {"human_node_rnn": h0}, masks, actions)
        loss = (v * wv).sum() / (T * N) + (lp * wl).sum() / (T * N) + 0.1 * ent
        pol.zero_grad()
        loss.backward()
        return v.detach(), lp.detach(), float(ent), hx["human_node_rnn"].detach(), {k: p.grad.detach().clone() for k, p in pol.named_parameters() if p.grad is not None}

    va, la, ea, ha, ga = run(a)
    vb, lb, eb, hb, gb = run(b)
    assert torch.allclose(va, vb, atol=2e-5, rtol=1e-5) and torch.allclose(la, lb, atol=5e-5, rtol=1e-5) and abs(ea - eb) < 1e-6
    assert torch.allclose(ha, hb, atol=2e-5)
    assert set(ga) == set(gb)
    for k in ga:
        scale = max(float(gb[k].abs().max()), 1e-4)
        err = float((ga[k] - gb[k]).abs().max())
        assert err <= 2e-4 * scale + 1e-7, (k, err, scale)


def test_fused_robot_node_sequence_head_gradients_at_a_few_samples():
    """T * N = 10 samples: every weight-gradient product of cn_rn_seq_bwd runs as ONE split, the smallest its scratch carve-up gets.  The
    reduced head gradients (776 floats) used to run past a 512-float region into the partial rows their own reduction was still reading:
    a race that showed as a handful of wrong dist.fc_mean.weight gradients once in a few runs.  100 backward passes must agree bit for bit
    with each other, and with the per-op module path."""
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    from tests import policy_util as PU
    torch.manual_seed(3)
    T, N, H, D = 5, 2, 5, 2
    ob_space, act_space = make_spaces(H, D)
    pol = Policy(ob_space.spaces, act_space, base="selfAttn_merge_srnn", base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=N, num_mini_batch=1, seq_length=T)).cuda()
    pol.base.train_fused_rn = True
    obs = {k: torch.from_numpy(v).cuda() for k, v in PU.synth_obs(T * N, H, D, seed=9).items()}
    g = torch.Generator(device="cuda").manual_seed(4)
    h0 = torch.randn(N, 1, 128, device="cuda", generator=g)
    masks = torch.ones(T * N, 1, device="cuda")
    actions = torch.randn(T * N, 2, device="cuda", generator=g)
    wv = torch.randn(T * N, 1, device="cuda", generator=g)
    wl = torch.randn(T * N, 1, device="cuda", generator=g)
    keys = ("dist.fc_mean.weight", "dist.fc_mean.bias", "base.critic_linear.weight", "base.critic_linear.bias", "dist.logstd._bias")
    first = None
    for _ in range(100):
        v, lp, _, _ = pol.evaluate_actions(obs, {"human_node_rnn": h0}, masks, actions)
        pol.zero_grad()
        ((v * wv).sum() + (lp * wl).sum()).backward()
        grads = {k: p.grad.detach().clone() for k, p in pol.named_parameters() if k in keys}
        if first is None:
            first = grads
        for k in keys:
            assert torch.equal(grads[k], first[k]), k
    # ... and with the per-op module path (autograd over the torch heads) on the same weights
    pol.base.train_fused_rn = False
    v, lp, _, _ = pol.evaluate_actions(obs, {"human_node_rnn": h0}, masks, actions)
    pol.zero_grad()
    ((v * wv).sum() + (lp * wl).sum()).backward()
    ref = {k: p.grad.detach().clone() for k, p in pol.named_parameters() if k in keys}
    for k in keys:
        assert float((first[k] - ref[k]).abs().max()) <= 2e-4 * max(float(ref[k].abs().max()), 1e-4) + 1e-7, k
