import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from crowdnav_prediction_attngraph_amd import _abi as A
from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch
import test_gpu_row_plan as T
for E, H, kw in [(4096, 20, {}), (1024, 12, {}), (4096, 5, {})]:
    env = HipEnvBatch(A.default_env_config(human_num=H, nenv=E, **kw), E, 425)
    obs = env.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(90):
        if t % 15 == 0:
            det = obs["detected_human_num"].view(E).cpu().numpy()
            loads, NW, n = T._check_plan(env.row_plan, det, E, H)
            d = np.clip(det.astype(int), 1, H)
            hist = np.bincount(d, minlength=H + 1)[1:]
            print(E, H, "t", t, "total", int(d.sum()), "tiles", len(loads), "n", n, "loads min/mean/max %d %.1f %d" % (loads.min(), loads.mean(), loads.max()), "blocks/WG max", int(((loads + 15) // 16).reshape(n, NW).sum(0).max()), "hist", hist.tolist())
        rn = obs["robot_node"].view(E, 7); gv = rn[:, 3:5] - rn[:, 0:2]
        a = 0.8 * gv / gv.norm(dim=1, keepdim=True).clamp_min(1e-6) + 0.3 * torch.randn(E, 2, device="cuda", generator=g)
        obs = env.step(a.contiguous())[0]
    env.close()
