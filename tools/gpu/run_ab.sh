#!/bin/bash
# first-fit plan builders (8 wavefronts) + pair-spreading lane kernel (F = this tree) against the previous build (H): whole GPU suite on F, A/B of the rollout
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/ab/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/ab/pytest.log
B="timeout 300 python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-ppo"
for v in H F H F; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so $B --timeline-out gpurun_out/ab/tl_$v.txt 2>> gpurun_out/ab/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v 200 steps', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launch_ms'))"
done
D="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --no-pmc-traffic --no-ppo"
for v in H F H F; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so $D 2>> gpurun_out/ab/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v driver form', d['value'], d['ms_per_step'], d['roofline']['frac'], 'worst', (d.get('worst_case') or {}).get('ms_per_step'))"
done
grep -v amdgpu.ids gpurun_out/ab/err.log | tail -3
