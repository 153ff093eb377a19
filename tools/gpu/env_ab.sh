#!/bin/bash
# the headline shape (4096 envs x 20 humans): the given "VAR=value" settings against the default, twice each
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-ppo --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs"
run() { env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('step_decomposition',{}).get('median_us'))"; }
for rep in 1 2; do
  run CN_X=0
  for a in "$@"; do run "$a"; done
done
