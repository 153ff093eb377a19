#!/bin/bash
# register-resident MT19937 in the pre-generation kernel (no LDS: the human-human kernel's workgroups get their CUs at once):
# H = shipped, HP = shipped + that, P = that + first-fit plan builders + pair-spreading lane kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/z
timeout 900 python -m pytest tests/test_gpu_env.py tests/test_gpu_row_plan.py -x -q -m gpu 2>&1 | tail -3
B="timeout 300 python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-ppo"
for v in H HP P H HP P; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so $B --timeline-out gpurun_out/z/tl_$v.txt 2>> gpurun_out/z/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launch_ms'))"
done
for v in H HP P; do echo $v; grep "^#   step" gpurun_out/z/tl_$v.txt | head -4 | cut -c1-200; done
grep -v amdgpu.ids gpurun_out/z/err.log | tail -3
