#!/bin/bash
# A/B of two builds of the library inside one box: bench.py (no PPO / CPU legs) alternately with the shipped library and with the ones given
# as arguments (paths relative to the repo root; CN_HIP_LIB), twice each.   gpurun -- 'bash tools/gpu/lib_ab.sh tools/gpu/_ab/lib_x.so'
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-ppo --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs"
run() { env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('step_decomposition',{}).get('median_us'))"; }
for rep in 1 2; do
  run CN_X=0
  for a in "$@"; do run "CN_HIP_LIB=$GRAFT_REPO_ROOT/$a"; done
done
