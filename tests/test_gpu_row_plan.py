"""-m gpu: the row plan the simulator writes beside every observation (csrc/row_plan.h) and the policy's use of it.

The plan is derived data of detected_human_num: the row offsets of the compacted (env, human) rows and a packing of the envs into equally
filled tiles for the fused human-human kernel.  Checked here: its invariants on real observations, that the forward with the plan equals
the forward without it (same rows, different tile composition: only the order of a few fp32 sums changes), and that configurations /
moments without a plan say so in the header instead of leaving a stale one behind."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

MAGIC = 0x52504C4E
TMAX = 1024


def _decode(plan, E):
    p = plan.cpu().numpy()
    hdr = p[:8]
    ro = p[8:8 + E + 1]
    oc = 8 + ((E + 1 + 3) & ~3)
    oi = oc + TMAX
    return hdr, ro, p[oc:oc + TMAX], p[oi:oi + TMAX * 64].reshape(TMAX, 64)


def _check_plan(plan, det, E, H):
    hdr, ro, cnt, items = _decode(plan, E)
    assert hdr[0] == MAGIC and hdr[4] == E and hdr[5] == H
    NW, n, total, T = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[6])
    rows_of = np.clip(det.astype(np.int64), 1, H)
    assert (ro == np.concatenate([[0], np.cumsum(rows_of)])).all() and total == rows_of.sum()
    assert NW == min(256, max(1, (E * H + 15) // 16)) and T == n * NW and n == max(1, -(-total // (62 * NW)))
    seen = np.zeros(E, int)
    loads = np.zeros(T, int)
    for b in range(T):
        ids, rows = items[b, :cnt[b]] & 0xFFFF, items[b, :cnt[b]] >> 16
        assert (rows == rows_of[ids]).all()
        seen[ids] += 1
        loads[b] = rows.sum()
    assert (seen == 1).all(), "every env in exactly one tile"
    assert loads.max() <= 63
    return loads, NW, n


@pytest.mark.parametrize("E,H,kw", [(4096, 20, {}), (4096, 20, dict(randomize_attributes=1, random_goal_changing=1)), (1024, 12, {}),
                                    (52, 5, {}), (4096, 5, {}), (512, 32, dict(circle_radius=12.0))])
def test_plan_invariants_and_balance(E, H, kw):
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
    env = HipEnvBatch(A.default_env_config(human_num=H, nenv=E, **kw), E, 425)
    obs = env.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    worst = 0
    for t in range(90):
        if t % 15 == 0:
            loads, NW, n = _check_plan(env.row_plan, obs["detected_human_num"].view(E).cpu().numpy(), E, H)
            # equally filled: no tile more than a few rows above the mean (what decides the 16-row blocks of the slowest workgroup).
            # Only where a tile holds enough envs to be levelled with: two envs of up to 32 rows per tile cannot be.
            if E >= 4 * len(loads):
                worst = max(worst, int(loads.max() - np.ceil(loads.mean())))
        rn = obs["robot_node"].view(E, 7)
        gv = rn[:, 3:5] - rn[:, 0:2]
        a = 0.8 * gv / gv.norm(dim=1, keepdim=True).clamp_min(1e-6) + 0.3 * torch.randn(E, 2, device="cuda", generator=g)
        obs = env.step(a.contiguous())[0]
    assert worst <= 2, worst
    env.close()


def test_planned_forward_equals_the_unplanned_one_and_fills_the_taps():
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch, HipPolicy
    from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
    E, H = 2048, 20
    env = HipEnvBatch(A.default_env_config(human_num=H, nenv=E), E, 425)
    torch.manual_seed(5)
    ob_space, act_space = make_spaces(H, 2)
    net = Policy(ob_space.spaces, act_space, base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=E), base="selfAttn_merge_srnn").cuda()
    pol = HipPolicy(H, 2, E)
    pol.set_weights(net.state_dict())
    obs = env.reset()
    h = torch.zeros(E, 1, 128, device="cuda")
    m = torch.ones(E, 1, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(60):
        eps = torch.randn(E, 2, device="cuda", generator=g)
        a = {k: v.clone() for k, v in pol.act(obs, h, m, eps=eps).items()}
        ta = pol.taps(E)
        b = pol.act(obs, h, m, eps=eps, row_plan=env.row_plan)
        tb = pol.taps(E)
        for k in a:
            assert float((a[k] - b[k]).abs().max()) <= 2e-5, (k, t)
        live = torch.arange(H, device="cuda").view(1, H) < obs["detected_human_num"].view(E, 1).clamp(1, H)
        assert float(((ta["spatial_lin"] - tb["spatial_lin"]).abs() * live.unsqueeze(-1)).max()) <= 4e-5
        obs, _, d, _, _, _ = env.step(b["action"].clone())
        h = b["hxs"].clone()
        m = (d == 0).float().view(E, 1)
    env.close()


def test_no_plan_is_said_in_the_header():
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
    # social-force humans: the step has no lane kernel to host the builder; E % 4 != 0: the builder's vector loads do not apply
    for E, kw in ((64, dict(human_num=10, humans_policy=1)), (50, dict(human_num=10)), (64, dict(human_num=40, circle_radius=14.0))):
        env = HipEnvBatch(A.default_env_config(nenv=E, **kw), E, 425)
        env.row_plan[:8].fill_(MAGIC)     # whatever was there before must not survive
        env.reset()
        assert int(env.row_plan[0]) == 0
        env.row_plan[:8].fill_(MAGIC)
        env.step(torch.zeros(E, 2, device="cuda"))
        assert int(env.row_plan[0]) == 0
        env.close()
    # a restored snapshot belongs to another observation than the plan in the buffer
    env = HipEnvBatch(A.default_env_config(human_num=10, nenv=64), 64, 425)
    env.reset()
    assert int(env.row_plan[0]) == MAGIC
    env.load_state_dict(env.state_dict())
    assert int(env.row_plan[0]) == 0
    env.close()


@pytest.mark.parametrize("E", [512, 1024, 2048, 4096])
def test_plan_buffer_of_arbitrary_content_is_filled_correctly(E):
    """The C ABI does not ask for a zeroed cn_obs.row_plan: a caller's hipMalloc'd buffer holds anything.  The grouped builder (several
    wavefronts at >= 1024 envs) keeps its arrival counter in the library's own memory, so a 0xFF-filled buffer -- header word 7 included --
    gets a complete, valid plan on every step (round 5 counted arrivals in header word 7 and relied on the binding's torch.zeros)."""
    from crowdnav_prediction_attngraph_amd import _abi as A
    from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
    H = 20
    env = HipEnvBatch(A.default_env_config(human_num=H, nenv=E), E, 425)
    env.row_plan.fill_(-1)                      # 0xFFFFFFFF everywhere
    obs = env.reset()
    _check_plan(env.row_plan, obs["detected_human_num"].view(E).cpu().numpy(), E, H)
    g = torch.Generator(device="cuda").manual_seed(7)
    for t in range(12):
        if t % 4 == 0:
            env.row_plan.fill_(-1)
        obs = env.step(torch.randn(E, 2, device="cuda", generator=g))[0]
        _check_plan(env.row_plan, obs["detected_human_num"].view(E).cpu().numpy(), E, H)
    env.close()
