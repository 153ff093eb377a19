#!/bin/bash
# valid A/B (CN_HIP_LIB honoured again): A = shipped kernels, B = weight-gradient kernel with buffer addressing + 128 x 512 tiles, C = nt stores in the training forward
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/u
export PYTHONPATH=$GRAFT_REPO_ROOT
for v in A B A B; do
  echo "== tn_bench lib$v"; CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so timeout 200 python tools/tn_bench.py 2>&1 | grep -v amdgpu.ids
done
echo "== tn_bench libB CN_TN_NB=2"; CN_TN_NB=2 CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libB.so timeout 200 python tools/tn_bench.py --shapes 1536x512 2>&1 | grep -v amdgpu.ids
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
for v in A B C A B C; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so $B 2>> gpurun_out/u/err.log | python -c "
import json,sys
from crowdnav_prediction_attngraph_amd import _abi
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppo']; print('$v', _abi.LIB_PATH[-8:], p.get('samples_per_s'), 'update_s', p.get('update_s'), 'rollout_s', p.get('rollout_s'), 'ms/step', d['ms_per_step'], p.get('error'))"
done
grep -v amdgpu.ids gpurun_out/u/err.log | tail -3
