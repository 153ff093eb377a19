#!/bin/bash
# A/B: cache policy of the training forward's activation stores (hh_fused_kernel<true>)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/i
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
for v in base nt base nt; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libcrowdnav_hip_$v.so $B 2>> gpurun_out/i/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppo']; print('$v', p.get('samples_per_s'), 'update_s', p.get('update_s'), 'rollout_s', p.get('rollout_s'), p.get('error'))"
done
grep -v amdgpu.ids gpurun_out/i/err.log | tail -3
