#!/bin/bash
# configs[4] (50 randomised humans x 8192 envs): time budget of one episode pre-generation launch (us) against the step and its kernels
cd $GRAFT_REPO_ROOT
B="python bench.py --humans 50 --randomized --envs 8192 --steps 40 --warmup 10 --dephase 120 --no-cpu-baseline --no-ppo --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs"
for a in "$@"; do
  $B --pregen-budget-us $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('budget $a us:', d['value'], d['ms_per_step'], d.get('step_decomposition',{}).get('median_us'))"
done
