#!/bin/bash
# configs[4] (50 randomised humans x 8192 envs): placement loops on one wavefront per env (CN_ENV_COOP=0), on four inside the step kernel
# (CN_ENV_DEFER=0), and deferred to env_post_kernel on the side stream (default); further "VAR=value" settings as arguments
cd $GRAFT_REPO_ROOT
B="python bench.py --humans 50 --randomized --envs 8192 --steps 40 --warmup 10 --dephase 120 --no-cpu-baseline --no-ppo --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs"
run() { env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('step_decomposition',{}).get('median_us'))"; }
run CN_ENV_COOP=0
run CN_ENV_DEFER=0
run CN_ENV_DEFER=1
for a in "$@"; do run "$a"; done
