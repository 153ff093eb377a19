#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w
B="timeout 300 python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-ppo"
for v in F H; do for bud in 55 40 30 20; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so $B --pregen-budget-us $bud 2>> gpurun_out/w/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v budget $bud', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launch_ms'))"
done; done
grep -v amdgpu.ids gpurun_out/w/err.log | tail -3
