#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/x
B="timeout 300 python bench.py --gpus 1 --steps 100 --warmup 30 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-ppo"
for v in F2; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so $B --timeline-out gpurun_out/x/tl_$v.txt 2>> gpurun_out/x/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  grep "^#   step" gpurun_out/x/tl_$v.txt | head -12
done
grep -v amdgpu.ids gpurun_out/x/err.log | tail -3
