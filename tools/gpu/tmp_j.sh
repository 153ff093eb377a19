#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_scale.py tests/test_gpu_ppo.py -m gpu -x -q > $O/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -4 $O/pytest1.log
cd /tmp; rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o u -- python $GRAFT_REPO_ROOT/examples/train_ppo.py --updates 4 > $O/update.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python examples/train_ppo.py --updates 4   (4 rollouts of 30 steps + 4 PPO updates = 40 optimiser steps, E=4096, H=20)" > $O/update_kernel_trace.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -c "Cijk\|rocblas" $O/update_kernel_trace.txt; grep "small_mm\|Cijk\|rocblas" $O/update_kernel_trace.txt | cut -c1-150 | head
tail -5 $O/update.log | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --no-pmc-traffic --no-worst-case --no-other-configs > $O/ppo.json 2> $O/ppo.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5j/ppo.json").read().strip().splitlines()[-1])
print("step", d["value"], d["ms_per_step"], "ppo", d["ppo"]["samples_per_s"], d["ppo"]["update_s"], d["ppo"]["rollout_s"])
PY
