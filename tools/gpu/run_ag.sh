#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ag
timeout 300 env CN_ATT8_MFMA=1 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "attention" 2>&1 | tail -2
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
for v in 0 1 0 1; do
  CN_ATT8_MFMA=$v $B 2>> gpurun_out/ag/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['ppo']; print('att8 mfma $v', p.get('samples_per_s'), 'update_s', p.get('update_s'), p.get('error'))"
done
grep -v amdgpu.ids gpurun_out/ag/err.log | tail -3
