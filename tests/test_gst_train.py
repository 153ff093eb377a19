"""GST training path vs the reference (tests/golden/gst_train_h20.npz, made by tests/golden/make_golden_gst_train.py from the
reference's TrajectoriesDataset / st_model / negative_log_likelihood_full_partial on a file of the data-collection env)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tests import policy_util  # noqa: F401,E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ITEMS = (0, 41, 77)
NAMES = ("obs_traj", "pred_traj", "obs_traj_rel", "pred_traj_rel", "loss_mask_rel", "loss_mask", "v_obs", "A_obs", "v_pred", "A_pred",
         "attn_mask_obs", "attn_mask_pred")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "gst_train_h20.npz"))


@pytest.fixture(scope="module")
def dataset(gold, tmp_path_factory):
    from crowdnav_prediction_attngraph_amd.gst_train import TrajectoriesDataset
    d = tmp_path_factory.mktemp("gstds")
    with open(str(d / "0.txt"), "w") as f:
        f.write(str(gold["file_lines"]) + "\n")
    return TrajectoriesDataset(str(d))


def _model(gold, prefix="w0_"):
    from crowdnav_prediction_attngraph_amd.gst import GSTPredictor
    m = GSTPredictor()
    m.load_state_dict({k[len(prefix):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(prefix)})
    return m


def test_dataset_equals_the_reference_class(gold, dataset):
    assert len(dataset) == int(gold["num_seq"]) == 111
    np.testing.assert_array_equal(np.array(dataset.seq_start_end), gold["seq_start_end"])
    np.testing.assert_array_equal(np.array(dataset.frame_id_seq), gold["frame_id_seq"])
    assert float(dataset.obs_traj.double().sum()) == float(gold["sum_obs_traj"]) and float(dataset.loss_mask_rel.sum()) == float(gold["sum_loss_mask_rel"])
    for it in ITEMS:
        for n, t in zip(NAMES, dataset[it]):
            np.testing.assert_array_equal(t.numpy(), gold["item%d_%s" % (it, n)], err_msg="item %d %s" % (it, n))


def test_training_forward_loss_and_gradients_match_the_reference(gold, dataset):
    from crowdnav_prediction_attngraph_amd import gst_train as T
    model = _model(gold)
    model.eval()
    for it in ITEMS:
        item = [t.unsqueeze(0) for t in dataset[it]]
        model.zero_grad()
        loss, gp, xs, info, v_pred_gt = T.sequence_loss(model, item, "cpu", 0.0)
        loss.backward()
        assert abs(loss.item() - float(gold["item%d_loss" % it])) <= 2e-5
        for n, t in zip(("mu", "sx", "sy", "corr"), gp):
            np.testing.assert_allclose(t.detach().numpy(), gold["item%d_%s" % (it, n)], rtol=0, atol=2e-5)
        lm = info["loss_mask_per_pedestrian"]
        np.testing.assert_allclose(T.average_offset_error(xs, v_pred_gt, lm).detach().numpy(), gold["item%d_aoe" % it], rtol=0, atol=2e-5)
        np.testing.assert_allclose(T.final_offset_error(xs, v_pred_gt, lm).detach().numpy(), gold["item%d_foe" % it], rtol=0, atol=2e-5)
        if it == ITEMS[0]:
            for k, p in model.named_parameters():
                ref = gold["grad0_" + k]
                assert float(np.abs(p.grad.numpy() - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max())), k


def test_six_optimiser_steps_match_the_reference_loop(gold, dataset):
    """train.py:113-149 on sequences 0..5 in order (Adam 1e-3, clip_grad 10, dropout off): losses and every weight afterwards."""
    from crowdnav_prediction_attngraph_amd import gst_train as T
    model = _model(gold)
    model.eval()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for it in range(6):
        loss = T.sequence_loss(model, [t.unsqueeze(0) for t in dataset[it]], "cpu", 0.0)[0]
        losses.append(loss.item())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        opt.zero_grad()
    np.testing.assert_allclose(losses, gold["sgd_losses"], rtol=0, atol=5e-5)
    for k, v in model.state_dict().items():
        assert float((v.numpy() - gold["after6_" + k]).__abs__().max()) <= 5e-5, k


def test_train_loop_learns_and_writes_a_loadable_checkpoint(gold, tmp_path):
    from crowdnav_prediction_attngraph_amd import gst_train as T
    from crowdnav_prediction_attngraph_amd.gst import GSTPredictor
    d = tmp_path / "data"
    d.mkdir()
    with open(str(d / "0.txt"), "w") as f:
        f.write(str(gold["file_lines"]) + "\n")
    model, hist = T.train(str(d), str(tmp_path / "run"), num_epochs=3, temp_epochs=4, save_epochs=2, device="cpu", log=lambda s: None)
    assert hist["epoch"] == 3 and hist["train_loss_task"][-1] < hist["train_loss_task"][0] and np.isfinite(hist["val_loss_task"]).all()
    m2 = GSTPredictor.from_checkpoint(str(tmp_path / "run" / "checkpoint" / "epoch_3.pt"), "cpu")
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_trained_run_directory_is_a_valid_model_dir(gold, tmp_path):
    """A run of gst_train.train() with num_epochs != 100 plugs back in as config.pred.model_dir: load_predictor picks
    'epoch_<num_epochs>.pt' from the run's own args (the reference loader's rule, crowd_nav_interface_multi_env_parallel.py:21-28),
    and the args.pickle / train_hist.pickle the reference's loaders open are there and hold plain argparse / dict objects."""
    import argparse
    import pickle
    import shutil
    from types import SimpleNamespace
    from crowdnav_prediction_attngraph_amd import gst_train as T
    from crowdnav_prediction_attngraph_amd.gst import find_checkpoint, load_predictor
    d = tmp_path / "data"
    d.mkdir()
    with open(str(d / "0.txt"), "w") as f:
        f.write(str(gold["file_lines"]) + "\n")
    run = tmp_path / "run"
    model, _ = T.train(str(d), str(run), num_epochs=5, temp_epochs=4, save_epochs=2, device="cpu", log=lambda s: None)
    ck = run / "checkpoint"
    assert sorted(p.name for p in ck.iterdir()) == ["args.json", "args.pickle", "epoch_2.pt", "epoch_4.pt", "epoch_5.pt", "train_hist.json", "train_hist.pickle"]
    with open(str(ck / "args.pickle"), "rb") as f:
        args = pickle.load(f)
    assert isinstance(args, argparse.Namespace) and args.num_epochs == 5 and args.spatial == "gumbel_social_transformer" and args.temporal == "faster_lstm"
    assert (args.embedding_size, args.spatial_num_heads, args.spatial_num_layers, args.spatial_num_heads_edges, args.lstm_hidden_size,
            args.obs_seq_len, args.pred_seq_len, args.output_dim, args.ghost) == (64, 8, 1, 0, 64, 5, 5, 5, False)
    with open(str(ck / "train_hist.pickle"), "rb") as f:
        assert pickle.load(f)["epoch"] == 5
    cfg = SimpleNamespace(pred=SimpleNamespace(model_dir=str(run)))
    assert find_checkpoint(str(run)).endswith("epoch_5.pt")
    loaded = load_predictor(cfg, "cpu")
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert torch.equal(a, b), k
    # reference-style directory (args.pickle only) and a bare directory of checkpoints
    (ck / "args.json").unlink()
    assert find_checkpoint(str(run)).endswith("epoch_5.pt")
    (ck / "args.pickle").unlink()
    shutil.copy(str(ck / "epoch_5.pt"), str(ck / "epoch_100.pt"))
    assert find_checkpoint(str(run)).endswith("epoch_100.pt")
