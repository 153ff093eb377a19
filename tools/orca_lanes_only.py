#!/usr/bin/env python3
"""Env-only stepping WITHOUT a row-plan buffer (MI355X): `orca_lane_kernel` then runs the ORCA agents alone, without the plan-building
workgroup -- the way to time the two halves of that launch separately (profiles/r03_orca_lane_variants.txt).

    rocprofv3 --kernel-trace --stats -d /tmp/lo -o t -- python tools/orca_lanes_only.py
    python profiles/summarize.py $(find /tmp/lo -name "*.db") "agents alone"
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crowdnav_prediction_attngraph_amd import _abi as A              # noqa: E402
from crowdnav_prediction_attngraph_amd.hip import HipEnvBatch        # noqa: E402

E, H = 4096, 20
env = HipEnvBatch(A.default_env_config(human_num=H, nenv=E), E, 425)
env.row_plan = None          # cn_obs.row_plan = NULL: no plan is built
env.reset()
g = torch.Generator(device="cuda")
g.manual_seed(1)
for t in range(400):
    env.step(torch.rand(E, 2, device="cuda", generator=g) * 2 - 1)
torch.cuda.synchronize()
print("done")
