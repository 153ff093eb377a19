#!/bin/bash
# robot-node sequence on the bf16x3 NT kernel: tests, then A/B of the PPO leg against the previous build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ppo.py -x -q -m gpu 2>&1 | tail -6
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
run() { echo "== $*"; env "$@" $B 2>> gpurun_out/p/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppo']; print(p.get('samples_per_s'), 'update_s', p.get('update_s'), 'rollout_s', p.get('rollout_s'), p.get('error'))"; }
run A=1
run CN_TRAIN_FUSED_RN=0
run A=1
run CN_TRAIN_FUSED_RN=0
grep -v amdgpu.ids gpurun_out/p/err.log | tail -3
