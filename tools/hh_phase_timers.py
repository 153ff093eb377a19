"""In-kernel phase timers of hh_fused_kernel (two-team schedule), run on the GPU box.

Needs a library built with the timers compiled in (they cost ~10 % of the kernel, so the production build has none):

    make -C crowdnav_prediction_attngraph_amd/csrc -B build/hh_fused.o HHFLAGS=-DHH_TIMING && make -C crowdnav_prediction_attngraph_amd/csrc
    BF_STEPS=240 python tools/hh_phase_timers.py
    make -C crowdnav_prediction_attngraph_amd/csrc -B build/hh_fused.o && make -C crowdnav_prediction_attngraph_amd/csrc     # back to production

Prints, per team (wavefront 0 of each), the s_memtime cycles per launch spent in each phase of the tile body, the launch time from HIP
events, the effective shader clock (slowest workgroup's cycles / launch time), and the workgroups grouped by their 16-row blocks per
launch -- the table behind "42 k cycles per tile + 27 k per row block" in DESIGN.md section 4.  The observation batch is a real one:
BF_STEPS simulator steps (default 240, like the bench's de-phasing pre-roll) under the sampled policy."""
import ctypes as C

import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crowdnav_prediction_attngraph_amd import _abi as A
from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch, HipPolicy
from crowdnav_prediction_attngraph_amd.policy import Policy, make_spaces
mode = "fused"
E, H = 4096, 20
env = HipEnvBatch(A.default_env_config(human_num=H, nenv=E), E, 425)
torch.manual_seed(425)
ob_space, act_space = make_spaces(H, 2)
net = Policy(ob_space.spaces, act_space, base_kwargs=dict(env_name="CrowdSimVarNum-v0", num_processes=E), base="selfAttn_merge_srnn").cuda()
pol = HipPolicy(H, 2, E); pol.set_gemm_mode(mode); pol.set_weights(net.state_dict())
obs = env.reset()
h = torch.zeros(E, 1, 128, device="cuda"); m = torch.ones(E, 1, device="cuda")
g = torch.Generator(device="cuda").manual_seed(1)
for t in range(int(os.environ.get("BF_STEPS", "40"))):   # get a realistic mid-episode observation set
    out = pol.act(obs, h, m, eps=torch.randn(E, 2, device="cuda", generator=g), row_plan=env.row_plan)
    obs, _, d, _, _, _ = env.step(out["action"].clone())
    h = out["hxs"].clone(); m = (d == 0).float().view(E, 1)
obs = {k: v.clone() for k, v in obs.items()}
plan = None if os.environ.get("HH_NO_PLAN") else env.row_plan.clone()   # the row plan made with this observation (HH_NO_PLAN=1: the kernel's own splitter)
torch.cuda.synchronize()
env.close()
torch.cuda.synchronize()
eps = torch.randn(E, 2, device="cuda", generator=g)
for _ in range(20):
    pol.act(obs, h, m, eps=eps, row_plan=plan)
torch.cuda.synchronize()
L = A.lib()
if not hasattr(L, "cn_hh_fused_set_timing"):
    raise SystemExit("this library was built without -DHH_TIMING (see the docstring)")
buf = torch.zeros(512 * 20, dtype=torch.int64, device="cuda")
L.cn_hh_fused_set_timing.argtypes = [C.c_void_p]
L.cn_hh_fused_set_timing(C.c_void_p(buf.data_ptr()))
N = 20
pol.set_profiling(True)
for _ in range(N):
    pol.act(obs, h, m, eps=eps, row_plan=plan)
torch.cuda.synchronize()
kms, kn = pol.get_profile()
print("hh kernel (timing build) mean %.1f us over %d launches" % (kms[0] / max(kn[0], 1) * 1e3, kn[0]))
L.cn_hh_fused_set_timing(C.c_void_p(0))
raw = buf.cpu().numpy().astype(np.float64) / N
b = raw.reshape(256, 2, 20)
names = ["e0+bar", "emb2+bar", "qkvloop", "bias+barA", "QKwrite+barB", "wos+S+softmax", "barC", "Pwrite+barD", "PV+Owrite", "barE", "os", "endbar", "exchange+finish"]
span_c, span_r, pro_c = b[:, 0, 13], b[:, 0, 14], b[:, 0, 15]
ok = span_r > 0
clk = (span_c[ok] / (span_r[ok] * 10e-9)).mean() / 1e9
print("shader clock during the kernel: %.3f GHz (s_memtime cycles / s_memrealtime 100 MHz ticks, mean over workgroups); workgroup span mean %.1f us max %.1f us; "
      "row-offset prologue mean %.0f cycles (%.2f us)" % (clk, span_r[ok].mean() * 0.01, span_r[ok].max() * 0.01, pro_c[ok].mean(), pro_c[ok].mean() / clk / 1e3))
for tm in range(2):
    bt = b[:, tm]
    tot = bt[:, :13].sum(1)
    print("  => effective clock %.2f GHz (max-block cycles / kernel time)" % (bt[:, :13].sum(1).max() / (kms[0] / max(kn[0], 1) * 1e-3) / 1e9))
    print("team %d: tiles/launch mean %.2f; rows mean %.1f; rbs mean %.2f; cycles per launch mean %.0f max %.0f min %.0f" % (tm, bt[:, 16].mean(), bt[:, 17].mean(), bt[:, 18].mean(), tot.mean(), tot.max(), tot.min()))
    for k, n in enumerate(names):
        print("  %-16s mean %9.0f  (%.1f%%)" % (n, bt[:, k].mean(), 100 * bt[:, k].mean() / tot.mean()))
bt = b[:, 0]
tot = bt[:, :13].sum(1)
import collections
for r in sorted(set(bt[:, 18].round().astype(int))):
    sel = bt[:, 18].round().astype(int) == r
    print("blocks with %d row-blocks/launch: %3d  mean cycles %.0f  max %.0f  min %.0f (rows mean %.1f, qkv %.0f)" % (r, sel.sum(), tot[sel].mean(), tot[sel].max(), tot[sel].min(), bt[sel, 17].mean(), bt[sel, 2].mean()))
xcd = np.arange(256) % 8
print("by XCD (blockIdx % 8): mean", [int(tot[xcd == k].mean()) for k in range(8)], " max", [int(tot[xcd == k].max()) for k in range(8)])
order = np.argsort(-tot)[:10]
print("slowest blocks (id, cycles, tiles, rows, rbs):", [(int(i), int(tot[i]), int(bt[i, 16]), int(bt[i, 17]), int(bt[i, 18])) for i in order])
order = np.argsort(tot)[:6]
print("fastest blocks:", [(int(i), int(tot[i]), int(bt[i, 16]), int(bt[i, 17]), int(bt[i, 18])) for i in order])
