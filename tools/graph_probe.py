"""Probe (run on the GPU box): capture K rollout steps (policy forward -> simulator step, side-stream ORCA / pre-generation included,
closed with cn_env_join) in a HIP graph through torch.cuda.graph and compare its replay with eager launches of the same steps.

    python tools/graph_probe.py

Round-3 result on MI355X / ROCm 7.2 (4096 envs x 20 humans, 20 captured steps): eager 219 us/step, graph replay 282 us/step -- the
cross-stream fork / join edges cost more as graph nodes than as eager event waits, so the rollout loop stays eager."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
from crowdnav_prediction_attngraph_amd import _abi as A
from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch, HipPolicy
from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
E, H = 4096, 20
env = HipEnvBatch(A.default_env_config(human_num=H, nenv=E), E, 425)
torch.manual_seed(425)
ob_space, act_space = make_spaces(H, 2)
net = Policy(ob_space.spaces, act_space, base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=E), base="selfAttn_merge_srnn").cuda()
pol = HipPolicy(H, 2, E); pol.set_gemm_mode("fused"); pol.set_weights(net.state_dict())
obs = env.reset()
hxs = [torch.zeros(E, 1, 128, device="cuda"), torch.zeros(E, 1, 128, device="cuda")]
masks = [torch.ones(E, 1, device="cuda"), torch.ones(E, 1, device="cuda")]
out = dict(value=torch.empty(E, 1, device="cuda"), action=torch.empty(E, 2, device="cuda"), logp=torch.empty(E, 1, device="cuda"), hxs=hxs[1])
eps = torch.randn(E, 2, device="cuda")
def step(i):
    out["hxs"] = hxs[(i + 1) & 1]
    pol.act(obs, hxs[i & 1], masks[i & 1], eps=eps, out=out)
    env.step(out["action"], not_done=masks[(i + 1) & 1])
for i in range(240): step(i)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for i in range(K * 10): step(i)
torch.cuda.synchronize()
print("eager: %.1f us/step" % ((time.perf_counter() - t0) / (K * 10) * 1e6))
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(4): step(i)          # warm up on the capture stream
        torch.cuda.synchronize()
        g.capture_begin()
        for i in range(K): step(i)
        env.join()
        g.capture_end()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print("graph replay: %.1f us/step" % ((time.perf_counter() - t0) / (K * 10) * 1e6))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:500])
