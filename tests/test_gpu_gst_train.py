"""-m gpu: the GST predictor's TRAINING path on the device (training half).

Round 6: the training step is a hand-written HIP path (cn_gst_train_step: forward + negative log-likelihood + backward, one workgroup per
sequence; cn_adam_clip_step for clip + Adam).  It is pinned to the reference's own numbers (tests/golden/gst_train_h20.npz: loss, Gaussian
parameters, every gradient, six optimiser steps of the reference loop with dropout off), to torch autograd on batches / padded crowds /
partially present pedestrians, its dropout masks are checked for consistency between the forward and the reverse pass (directional
derivatives under a fixed seed) and a short training run through it learns and plugs into the HIP inference kernels.  The torch-op graph
of gst_train.py (the CPU tests' path and the cross-check of the kernels) is pinned on the device as before."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.test_gst_train import GOLDEN, ITEMS, _model  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "gst_train_h20.npz"))


@pytest.fixture(scope="module")
def dataset(gold, tmp_path_factory):
    from crowdnav_prediction_attngraph_amd.gst_train import TrajectoriesDataset
    d = tmp_path_factory.mktemp("gstds_gpu")
    with open(str(d / "0.txt"), "w") as f:
        f.write(str(gold["file_lines"]) + "\n")
    return TrajectoriesDataset(str(d))


def test_training_forward_loss_and_gradients_on_the_device_match_the_reference(gold, dataset):
    from crowdnav_prediction_attngraph_amd import gst_train as T
    model = _model(gold).cuda()
    model.eval()
    for it in ITEMS:
        item = [t.unsqueeze(0) for t in dataset[it]]
        model.zero_grad()
        loss, gp, xs, info, v_pred_gt = T.sequence_loss(model, item, "cuda", 0.0)
        loss.backward()
        assert abs(loss.item() - float(gold["item%d_loss" % it])) <= 5e-5
        for n, t in zip(("mu", "sx", "sy", "corr"), gp):
            np.testing.assert_allclose(t.detach().cpu().numpy(), gold["item%d_%s" % (it, n)], rtol=0, atol=5e-5)
        if it == ITEMS[0]:
            for k, p in model.named_parameters():
                ref = gold["grad0_" + k]
                assert float(np.abs(p.grad.cpu().numpy() - ref).max()) <= 5e-5 * max(1.0, float(np.abs(ref).max())), k


def test_six_optimiser_steps_on_the_device_and_the_hip_predictor_runs_the_result(gold, dataset):
    from crowdnav_prediction_attngraph_amd import gst_train as T
    from crowdnav_prediction_attngraph_amd.hip import HipGST
    model = _model(gold).cuda()
    model.eval()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for it in range(6):
        loss = T.sequence_loss(model, [t.unsqueeze(0) for t in dataset[it]], "cuda", 0.0)[0]
        losses.append(loss.item())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        opt.zero_grad()
    np.testing.assert_allclose(losses, gold["sgd_losses"], rtol=0, atol=1e-4)
    for k, v in model.state_dict().items():
        assert float((v.cpu().numpy() - gold["after6_" + k]).__abs__().max()) <= 1e-4, k
    # the trained weights go straight into the HIP inference kernels (cn_gst_set_weights / cn_gst_predict): same predictions as the torch module
    E, H = 3, 20
    g = torch.Generator().manual_seed(5)
    traj = torch.cumsum(0.2 * torch.randn(E, H, 5, 2, generator=g), 2).cuda()
    mask = (torch.rand(E, H, 5, generator=g) > 0.15).float().cuda()
    hip = HipGST(H, E)
    hip.set_weights(model.state_dict())
    out_traj, out_mask = hip.predict(traj, mask)
    with torch.no_grad():
        ref_traj, ref_mask = model(traj, mask.unsqueeze(-1))
    valid = ref_mask.bool().view(E, H)
    assert torch.equal(out_mask.view(E, H).bool(), valid)
    err = (out_traj.view(E, H, 5, 5)[valid] - ref_traj.view(E, H, 5, 5)[valid]).abs().max()
    assert float(err) <= 1e-4, float(err)


# ---- round 6: the HIP training step (cn_gst_train_step / HipGstTrainer) ----
def _hip_inputs(item):
    return item[6], item[8], item[4]          # v_obs [1,5,N,2], v_pred [1,5,N,2], loss_mask_rel [1,N,10]


def test_hip_training_step_matches_the_reference_loss_gaussians_and_gradients(gold, dataset):
    from crowdnav_prediction_attngraph_amd import gst_train as T
    model = _model(gold).cuda()
    tr = T.HipGstTrainer(model)
    for it in ITEMS:
        item = [t.unsqueeze(0) for t in dataset[it]]
        out, gauss = tr.loss_and_grads(*_hip_inputs(item), p_drop=0.0)
        assert abs(float(out[0]) - float(gold["item%d_loss" % it])) <= 2e-5
        g = gauss.cpu().numpy()
        for n, sl in (("mu", slice(0, 2)), ("sx", slice(2, 3)), ("sy", slice(3, 4)), ("corr", slice(4, 5))):
            np.testing.assert_allclose(g[..., sl], gold["item%d_%s" % (it, n)], rtol=0, atol=2e-5)
        if it == ITEMS[0]:
            for k, p in model.named_parameters():
                ref = gold["grad0_" + k]
                assert float(np.abs(p.grad.cpu().numpy() - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max())), k


def test_hip_six_optimiser_steps_match_the_reference_loop_and_feed_the_hip_predictor(gold, dataset):
    """train.py:113-149 on sequences 0..5 in order (Adam 1e-3, clip_grad 10, dropout off) entirely through the C ABI: cn_gst_train_step +
    cn_adam_clip_step; losses and every weight afterwards against the reference's; the result runs in the HIP inference kernels."""
    from crowdnav_prediction_attngraph_amd import gst_train as T
    from crowdnav_prediction_attngraph_amd.hip import HipGST
    model = _model(gold).cuda()
    tr = T.HipGstTrainer(model, lr=1e-3, clip_grad=10.0)
    losses = []
    for it in range(6):
        out, _ = tr.loss_and_grads(*_hip_inputs([t.unsqueeze(0) for t in dataset[it]]), p_drop=0.0)
        losses.append(float(out[0]))
        tr.optimizer_step()
    np.testing.assert_allclose(losses, gold["sgd_losses"], rtol=0, atol=5e-5)
    # (1e-4 like the torch-graph test above: the key bias of in_proj has a gradient of exactly zero in exact arithmetic -- softmax is shift
    # invariant -- so Adam normalises pure rounding noise there and the entries wander by up to lr x 6 steps x a fraction; measured 5.9e-5)
    for k, v in model.state_dict().items():
        assert float(np.abs(v.cpu().numpy() - gold["after6_" + k]).max()) <= 1e-4, k
    E, H = 3, 20
    g = torch.Generator().manual_seed(5)
    traj = torch.cumsum(0.2 * torch.randn(E, H, 5, 2, generator=g), 2).cuda()
    mask = (torch.rand(E, H, 5, generator=g) > 0.15).float().cuda()
    hg = HipGST(H, E)
    hg.set_weights(model.state_dict())
    out_traj, out_mask = hg.predict(traj, mask)
    with torch.no_grad():
        ref_traj, ref_mask = model(traj, mask.unsqueeze(-1))
    valid = ref_mask.bool().view(E, H)
    assert torch.equal(out_mask.view(E, H).bool(), valid)
    assert float((out_traj.view(E, H, 5, 5)[valid] - ref_traj.view(E, H, 5, 5)[valid]).abs().max()) <= 1e-4


@pytest.mark.parametrize("B,N,seed", [(1, 3, 0), (2, 20, 1), (3, 37, 2), (1, 64, 3)])
def test_hip_training_step_equals_torch_autograd_on_batches_and_ragged_presence(B, N, seed):
    """Random weights, B sequences of N pedestrians with pedestrians missing at some steps (and crowds below the kernel's minimum of four: padded):
    loss, Gaussian parameters and every gradient of the pooled loss (sum of masked NLL / valid pairs of the whole batch) vs torch autograd on
    the op graph of gst_train.py, dropout off.  Bars: 2e-5 on the loss / parameters, 1e-4 of a gradient tensor's largest entry."""
    from crowdnav_prediction_attngraph_amd import gst_train as T
    from crowdnav_prediction_attngraph_amd.gst import GSTPredictor
    torch.manual_seed(100 + seed)
    model = GSTPredictor().cuda()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    g = torch.Generator().manual_seed(seed)
    lm = (torch.rand(B, N, 10, generator=g) > 0.25).float()
    lm[:, 0] = 1.0                                            # one pedestrian present throughout (the dataset's rule)
    lm[:, 1, 4] = 0.0                                         # ... and one without a last observed step (not predicted at all)
    if N > 2:
        lm[:, 2, :] = 0.0                                     # ... and one never present
    v_obs = torch.where(lm[:, :, :5].permute(0, 2, 1).unsqueeze(-1) > 0, 0.4 * torch.randn(B, 5, N, 2, generator=g), torch.full((B, 5, N, 2), -999.0))
    v_pred = torch.where(lm[:, :, 5:].permute(0, 2, 1).unsqueeze(-1) > 0, 0.4 * torch.randn(B, 5, N, 2, generator=g), torch.full((B, 5, N, 2), -999.0))
    # reference: pooled loss over the batch
    model.zero_grad()
    num, den, gps = 0.0, 0.0, []
    for b in range(B):
        l1 = lm[b:b + 1].cuda()
        am = (l1[0].t().unsqueeze(2) * l1[0].t().unsqueeze(1))[:5].unsqueeze(0)
        gp, xs, info = T.forward_train(model, v_obs[b:b + 1].cuda(), am, l1, 0.0)
        pl, elm = T.negative_log_likelihood_full_partial(gp, v_pred[b:b + 1].cuda(), info["loss_mask_rel_full_partial"], l1[:, :, 5:])
        num, den = num + pl.sum(), den + elm.sum()
        gps.append(torch.cat(gp, -1))
    loss_ref = num / den
    loss_ref.backward()
    loss_ref, den = loss_ref.detach(), den.detach()
    ref_g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    ref_gauss = torch.cat(gps, 0).detach()
    tr = T.HipGstTrainer(model)                               # rebinds .grad to the bucket
    out, gauss = tr.loss_and_grads(v_obs, v_pred, lm, p_drop=0.0)
    assert abs(float(out[0]) - float(loss_ref)) <= 2e-5 * max(1.0, abs(float(loss_ref))) and abs(float(out[1]) - float(den)) < 0.5
    assert float((gauss - ref_gauss).abs().max()) <= 2e-5 * max(1.0, float(ref_gauss.abs().max()))
    for k, p in model.named_parameters():
        scale = max(float(ref_g[k].abs().max()), 1e-6)
        assert float((p.grad - ref_g[k]).abs().max()) <= 1e-4 * scale + 1e-7, (k, float((p.grad - ref_g[k]).abs().max()), scale)


def test_hip_dropout_masks_are_the_same_in_the_forward_and_the_reverse_pass():
    """With p_drop = 0.1 and a FIXED seed the loss is a smooth function of the weights (the masks do not move): its central difference along a
    random direction must equal gradient . direction -- which it only does if the reverse pass regenerates exactly the forward's masks at all
    four sites.  Also: same seed -> identical results, another seed -> another loss, and about 10 % of what a site produces is zeroed."""
    from crowdnav_prediction_attngraph_amd import gst_train as T
    from crowdnav_prediction_attngraph_amd.gst import GSTPredictor
    torch.manual_seed(7)
    model = GSTPredictor().cuda().double().float()
    g = torch.Generator().manual_seed(11)
    B, N = 2, 24
    lm = (torch.rand(B, N, 10, generator=g) > 0.15).float()
    lm[:, 0] = 1.0
    v_obs, v_pred = 0.4 * torch.randn(B, 5, N, 2, generator=g), 0.4 * torch.randn(B, 5, N, 2, generator=g)
    tr = T.HipGstTrainer(model)
    l0, _ = tr.loss_and_grads(v_obs, v_pred, lm, p_drop=0.1, seed=5)
    g0 = tr.flat["g"].clone()
    l0b, _ = tr.loss_and_grads(v_obs, v_pred, lm, p_drop=0.1, seed=5)
    assert torch.equal(l0, l0b) and torch.equal(g0, tr.flat["g"])
    l1, _ = tr.loss_and_grads(v_obs, v_pred, lm, p_drop=0.1, seed=6)
    lz, _ = tr.loss_and_grads(v_obs, v_pred, lm, p_drop=0.0, seed=5)
    assert float(l1[0]) != float(l0[0]) and float(lz[0]) != float(l0[0])
    d = torch.randn(tr.flat["p"].shape, generator=torch.Generator().manual_seed(2)).cuda()
    d /= d.norm()
    w0 = tr.flat["p"].clone()
    eps = 2e-2
    tr.flat["p"].copy_(w0 + eps * d)
    lp = float(tr.loss_and_grads(v_obs, v_pred, lm, p_drop=0.1, seed=5)[0][0])
    tr.flat["p"].copy_(w0 - eps * d)
    lmn = float(tr.loss_and_grads(v_obs, v_pred, lm, p_drop=0.1, seed=5)[0][0])
    tr.flat["p"].copy_(w0)
    fd, an = (lp - lmn) / (2 * eps), float((g0 * d).sum())
    assert abs(fd - an) <= 0.03 * max(abs(an), 1e-3) + 2e-4, (fd, an)


def test_training_run_through_the_hip_step_learns_and_writes_a_loadable_checkpoint(gold, tmp_path):
    from crowdnav_prediction_attngraph_amd import gst_train as T
    from crowdnav_prediction_attngraph_amd.gst import GSTPredictor
    d = tmp_path / "data"
    d.mkdir()
    with open(str(d / "0.txt"), "w") as f:
        f.write(str(gold["file_lines"]) + "\n")
    model, hist = T.train(str(d), str(tmp_path / "run"), num_epochs=3, temp_epochs=4, save_epochs=2, device="cuda", log=lambda s: None, backend="hip")
    assert hist["epoch"] == 3 and hist["train_loss_task"][-1] < hist["train_loss_task"][0] and np.isfinite(hist["val_loss_task"]).all()
    m2 = GSTPredictor.from_checkpoint(str(tmp_path / "run" / "checkpoint" / "epoch_3.pt"), "cuda")
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
