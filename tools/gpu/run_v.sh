#!/bin/bash
# first-fit plan builders (8 wavefronts) + pair-spreading lane kernel (F) against the shipped single water-filling builder (H)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v
python tools/row_plan_probe.py tools/det_counts_sample.npz 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python -m pytest tests/test_gpu_row_plan.py tests/test_gpu_env.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
B="timeout 300 python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-ppo"
for v in H F H F; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so $B --timeline-out gpurun_out/v/tl_$v.txt 2>> gpurun_out/v/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launch_ms'))"
done
grep -v amdgpu.ids gpurun_out/v/err.log | tail -3
