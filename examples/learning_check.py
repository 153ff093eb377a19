"""End-to-end learning check on one MI355X: PPO with the reference's hyper-parameters on the batched device simulator, then the
500-case deterministic evaluation.  usage: python examples/learning_check.py <lr> <envs> <updates>   (e.g. 4e-5 512 400)"""
import sys, json, time, torch
sys.path.insert(0, ".")
from crowdnav_prediction_attngraph_amd import config as C
from crowdnav_prediction_attngraph_amd.trainer import train
from crowdnav_prediction_attngraph_amd.evaluation import evaluate_batched
import logging
logging.basicConfig(level=logging.INFO, stream=sys.stdout)
lr, E, U = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = C.non_randomized(**{"sim.human_num": 20})
t0 = time.time()
acc = []
def log(r):
    acc.append(r)
    if (r["update"] + 1) % max(U // 12, 1) == 0:
        w = acc[-max(U // 12, 1):]
        ep = sum(x["episodes"] for x in w) or 1
        print("update %4d  steps %.1fM  eprewmean %7.2f  success %.2f collision %.2f timeout %.2f  entropy %.3f  (%.0f s)" % (
            r["update"] + 1, (r["update"] + 1) * 30 * E / 1e6, sum(x["eprewmean"] * x["episodes"] for x in w) / ep,
            sum(x["success"] * x["episodes"] for x in w) / ep, sum(x["collision"] * x["episodes"] for x in w) / ep,
            sum(x["timeout"] * x["episodes"] for x in w) / ep, r["entropy"], time.time() - t0), flush=True)
hist, pol = train("CrowdSimVarNum-v0", E, 30, U, 425, config=cfg, lr=lr, log=log)
print("training wall time %.1f s for %.1f M env steps" % (time.time() - t0, U * 30 * E / 1e6))
m = evaluate_batched(pol, "CrowdSimVarNum-v0", cfg, 425, 500, logging=logging.getLogger("eval"))
print({k: v for k, v in m.items() if "cases" not in k})
