"""Stand-alone timing of the weight-gradient product (cn_linear_wgrad) at the PPO update's shapes: M live rows of a minibatch,
dW[N][K] = dY^T X.  Prints ms per call (HIP events, median of --iters) and the bf16x3 MFMA rate.  GPU box only."""
import argparse
import statistics
import torch
from crowdnav_prediction_attngraph_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=358400)
ap.add_argument("--shapes", default="1536x512,512x512,512x256")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
torch.manual_seed(0)
for sh in a.shapes.split(","):
    N, K = (int(v) for v in sh.split("x"))
    dy = torch.randn(a.m, N, device="cuda")
    x = torch.randn(a.m, K, device="cuda")
    for _ in range(3):
        hip.wgrad(dy, x)
    ts = []
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.wgrad(dy, x); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = statistics.median(ts)
    print("wgrad M=%d N=%d K=%d: %.3f ms (min %.3f)  %.0f TFLOP/s bf16x3-executed" % (a.m, N, K, ms, min(ts), 6.0 * a.m * N * K / ms / 1e9))
    del dy, x
