#!/bin/bash
# configs[3] (CrowdSimPredRealGST-v0, 2048 envs x 20 humans): the given "VAR=value" settings against the default, twice
cd $GRAFT_REPO_ROOT
B="python bench.py --env-name CrowdSimPredRealGST-v0 --envs 2048 --steps 60 --warmup 20 --no-cpu-baseline --no-ppo --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs"
run() { env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  run CN_X=0
  for a in "$@"; do run "$a"; done
done
