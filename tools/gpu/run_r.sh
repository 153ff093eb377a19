#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ppo.py -x -q -m gpu 2>&1 | tail -6
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
run() { echo "== $*"; env "$@" $B 2>> gpurun_out/r/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppo']; print(p.get('samples_per_s'), 'update_s', p.get('update_s'), 'rollout_s', p.get('rollout_s'), p.get('error'))"; }
run A=1
run A=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o u -- python $GRAFT_REPO_ROOT/examples/train_ppo.py --updates 4 > $OUT/update.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python examples/train_ppo.py --updates 4   (4 rollouts of 30 steps + 4 PPO updates = 40 optimiser steps, E=4096, H=20)" > $OUT/update_kernel_trace.txt 2>&1
grep "gru_seq\|rn_reduce\|rn_rl" $OUT/update_kernel_trace.txt | cut -c1-175
grep -v amdgpu.ids $OUT/err.log | tail -3
