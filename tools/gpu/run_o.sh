#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/o
B="timeout 300 python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-ppo"
for v in default high default high; do
  $B --stream-priority $v --timeline-out gpurun_out/o/tl_$v.txt 2>> gpurun_out/o/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
# the driver's launch form for N > 1, with one rank (RCCL initialised)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic 2>> gpurun_out/o/err.log | tail -1 | cut -c1-300
grep -v amdgpu.ids gpurun_out/o/err.log | tail -3
