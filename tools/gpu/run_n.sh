#!/bin/bash
cd $GRAFT_REPO_ROOT
PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, time
from crowdnav_prediction_attngraph_amd import hip
M, N, K = 364000, 1536, 512
torch.manual_seed(0)
dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
for tag, scale in (("randn", 1.0), ("sparse-ish (relu of randn, 50% zeros)", None)):
    if scale is None:
        dy = torch.relu(dy); x = torch.relu(x)
    for _ in range(3): hip.wgrad(dy, x)
    torch.cuda.synchronize()
    # back-to-back launches without host syncs: 600 calls ~ 1 s of sustained load; time blocks of 50
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
    evs[0].record()
    for b in range(12):
        for _ in range(50): hip.wgrad(dy, x)
        evs[b + 1].record()
    torch.cuda.synchronize()
    print(tag, "ms per call in consecutive blocks of 50:", " ".join("%.3f" % (evs[b].elapsed_time(evs[b + 1]) / 50) for b in range(12)))
PY
