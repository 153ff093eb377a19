#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ah
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_dist.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
for v in 0 1 0 1; do
  CN_PPO_GATHER_GRADS=$v $B 2>> gpurun_out/ah/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppo']; print('gather $v', p.get('samples_per_s'), 'update_s', p.get('update_s'), p.get('error'))"
done
grep -v amdgpu.ids gpurun_out/ah/err.log | tail -3
