#!/bin/bash
# A/B: TN (weight-gradient) kernel tile width / split count via env knobs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
run() { echo "== $*"; env "$@" $B 2>> gpurun_out/k/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppo']; print(p.get('samples_per_s'), 'update_s', p.get('update_s'), 'rollout_s', p.get('rollout_s'), p.get('error'))"; }
run A=0
run CN_TN_NB=1
run CN_TN_NB=1 CN_TN_SPLITS=32
run CN_TN_NB=1 CN_TN_SPLITS=64
run CN_TN_SPLITS=64
run A=0
grep -v amdgpu.ids gpurun_out/k/err.log | tail -3
