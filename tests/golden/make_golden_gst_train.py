#!/usr/bin/env python3
"""Golden vectors for the GST TRAINING path (build container only): the reference's dataset class and training-time forward / loss,
fed with a file of the data-collection env.

  * gst_updated/src/mgnn/trajectories.py TrajectoriesDataset (obs 5 / pred 5, skip 1, frame_diff 1) over the first 120 frames of
    tests/golden/collect_h20_nonrand_r0.npz written the way collect_data.py writes them;
  * gst_updated/src/gumbel_social_transformer/st_model.py forward (tau 0.5, hard False, sampling False -- train.py:128) with formula
    weights in eval mode (dropout off), negative_log_likelihood_full_partial, the gradients of every parameter, aoe / foe;
  * six optimiser steps of train.py's inner loop (Adam lr 1e-3, clip_grad 10, sequences in order, no rotation, dropout off).
"""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_import as R  # noqa: E402
import make_golden_gst as MG  # noqa: E402

R.install()
sys.path.insert(0, os.path.join(R.REF, "gst_updated"))
FRAMES = 120
ITEMS = (0, 41, 77)


def main():
    import torch
    from src.mgnn.trajectories import TrajectoriesDataset
    from src.mgnn.utils import average_offset_error, final_offset_error
    from gst_updated.src.gumbel_social_transformer.st_model import st_model, negative_log_likelihood_full_partial
    z = np.load(os.path.join(HERE, "collect_h20_nonrand_r0.npz"))
    lines = [ln for ln in str(z["lines"]).split("\n") if float(ln.split("\t")[0]) < FRAMES]
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "0.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
        ds = TrajectoriesDataset(d, obs_seq_len=5, pred_seq_len=5, skip=1, delim="\t", frame_diff=1.0)
    out = {"num_seq": np.array(len(ds)), "seq_start_end": np.array(ds.seq_start_end), "frame_id_seq": np.array(ds.frame_id_seq),
           "sum_obs_traj": np.array(float(ds.obs_traj.double().sum())), "sum_loss_mask_rel": np.array(float(ds.loss_mask_rel.sum())),
           "file_lines": np.array("\n".join(lines))}
    names = ("obs_traj", "pred_traj", "obs_traj_rel", "pred_traj_rel", "loss_mask_rel", "loss_mask", "v_obs", "A_obs", "v_pred", "A_pred",
             "attn_mask_obs", "attn_mask_pred")
    for it in ITEMS:
        for n, t in zip(names, ds[it]):
            out["item%d_%s" % (it, n)] = t.numpy()
    args = argparse.Namespace(**MG.GST_ARGS)
    torch.manual_seed(0)
    model = st_model(args, device="cpu")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = MG.gst_formula_state_dict(shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    for k, v in sd.items():
        out["w0_" + k] = v

    def loss_of(item):
        obs_traj, pred_traj_gt, obs_traj_rel, pred_traj_rel_gt, loss_mask_rel, loss_mask, v_obs, A_obs, v_pred_gt, A_pred_gt, amo, amp = [t.unsqueeze(0) for t in item]
        gp, xs, info = model(v_obs, A_obs, amo, loss_mask_rel, tau=0.5, hard=False, sampling=False, device="cpu")
        lm_fp = info["loss_mask_rel_full_partial"]
        prob_loss, elm = negative_log_likelihood_full_partial(gp, v_pred_gt, lm_fp, loss_mask_rel[:, :, -args.pred_seq_len:])
        loss = prob_loss.sum() / elm.sum()
        return loss, gp, xs, info, v_pred_gt

    for it in ITEMS:
        model.zero_grad()
        loss, gp, xs, info, v_pred_gt = loss_of(ds[it])
        loss.backward()
        out["item%d_loss" % it] = np.array(loss.item())
        for n, t in zip(("mu", "sx", "sy", "corr"), gp):
            out["item%d_%s" % (it, n)] = t.detach().numpy()
        out["item%d_aoe" % it] = average_offset_error(xs, v_pred_gt, loss_mask=info["loss_mask_per_pedestrian"]).detach().numpy()
        out["item%d_foe" % it] = final_offset_error(xs, v_pred_gt, loss_mask=info["loss_mask_per_pedestrian"]).detach().numpy()
        if it == ITEMS[0]:
            for k, p in model.named_parameters():
                out["grad0_" + k] = p.grad.detach().numpy().copy()
    # six steps of train.py's inner loop (:113-149) on sequences 0..5, dropout off
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt.zero_grad()
    losses = []
    for it in range(6):
        loss = loss_of(ds[it])[0]
        losses.append(loss.item())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        opt.zero_grad()
    out["sgd_losses"] = np.array(losses)
    for k, v in model.state_dict().items():
        out["after6_" + k] = v.numpy().copy()
    path = os.path.join(HERE, "gst_train_h20.npz")
    np.savez_compressed(path, **out)
    print("sequences %d, items %s, losses %s -> %s (%.0f KB)" % (len(ds), ITEMS, [round(x, 4) for x in losses], os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
