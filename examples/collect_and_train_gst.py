#!/usr/bin/env python3
"""End to end, the way the reference's pipeline goes (collect_data.py -> gst_updated train.py -> config.pred.model_dir):
simulate crowds on the GPU, write the GST dataset files, train the predictor on them, load the checkpoint back.

    python examples/collect_and_train_gst.py [--envs 256] [--steps 400] [--epochs 5] [--out /tmp/gst_run]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crowdnav_prediction_attngraph_amd import config as C  # noqa: E402
from crowdnav_prediction_attngraph_amd import gst_train  # noqa: E402
from crowdnav_prediction_attngraph_amd.collect import CollectVecEnv, collect_lines  # noqa: E402
from crowdnav_prediction_attngraph_amd.gst import GSTPredictor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--train-files", type=int, default=4, help="how many of the env files the (per-sequence, host-driven) training loop reads")
    ap.add_argument("--out", default="/tmp/gst_run")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = C.non_randomized(**{"sim.human_num": 20, "robot.policy": "orca"})
    envs = CollectVecEnv(425, a.envs, dev, config=cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lines = collect_lines(envs, a.steps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    envs.close()
    data_dir = os.path.join(a.out, "data")
    os.makedirs(data_dir, exist_ok=True)
    rows = 0
    for i in range(min(a.train_files, a.envs)):
        with open(os.path.join(data_dir, "%d.txt" % i), "w") as f:
            f.write("\n".join(lines[i]) + "\n")
    rows = sum(len(x) for x in lines)
    print("collected %d envs x %d steps = %d env-steps, %d (frame, id, x, y) rows in %.2f s (%.0f env-steps/s incl. the text formatting on the host)"
          % (a.envs, a.steps, a.envs * a.steps, rows, t1 - t0, a.envs * a.steps / (t1 - t0)))
    model, hist = gst_train.train(data_dir, os.path.join(a.out, "run"), num_epochs=a.epochs, temp_epochs=max(a.epochs, 4), save_epochs=a.epochs, device=dev)
    ck = os.path.join(a.out, "run", "checkpoint", "epoch_%d.pt" % a.epochs)
    GSTPredictor.from_checkpoint(ck, dev)
    print("val aoe %.4f -> %.4f, val foe %.4f -> %.4f; checkpoint %s" % (hist["val_aoe_task"][0], hist["val_aoe_task"][-1], hist["val_foe_task"][0],
                                                                      hist["val_foe_task"][-1], ck))


if __name__ == "__main__":
    main()
