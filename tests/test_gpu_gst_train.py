"""-m gpu: the GST predictor's TRAINING path on the device (training half).  It is a torch-op graph (gst_train.py: a
sequence is a few dozen pedestrians x 10 frames, 67 k parameters; DESIGN.md 1 says why it has no hand-written kernels): this test pins that
graph ON THE MI355X to the reference's own numbers (tests/golden/gst_train_h20.npz: loss, Gaussian parameters, offset errors, every
gradient, six optimiser steps of the reference loop) and checks that what it trains is what the HIP inference kernels then run."""
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
