#!/bin/bash
# TN (weight-gradient) kernel: buffer addressing + 128 x 512 tiles; tests, then A/B of the PPO leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/l
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ppo.py -x -q -m gpu 2>&1 | tail -4
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
run() { echo "== $*"; env "$@" $B 2>> gpurun_out/l/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppo']; print(p.get('samples_per_s'), 'update_s', p.get('update_s'), 'rollout_s', p.get('rollout_s'), p.get('error'))"; }
run CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libcrowdnav_hip_base.so
run CN_TN_NB=2
run CN_TN_NB=4
run CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libcrowdnav_hip_base.so
run CN_TN_NB=2
run CN_TN_NB=4
grep -v amdgpu.ids gpurun_out/l/err.log | tail -3
