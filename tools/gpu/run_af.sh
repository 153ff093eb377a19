#!/bin/bash
# attention backward of the 9..16 class on the matrix pipe (M) against the VALU kernel (A)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/af
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ppo.py -x -q -m gpu 2>&1 | tail -4
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
for v in A M A M; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so $B 2>> gpurun_out/af/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppo']; print('$v', p.get('samples_per_s'), 'update_s', p.get('update_s'), 'rollout_s', p.get('rollout_s'), p.get('error'))"
done
grep -v amdgpu.ids gpurun_out/af/err.log | tail -3
