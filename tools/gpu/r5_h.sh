#!/bin/bash
# round 5, call H: packed point lists in the placement screen (bit-exact suites, configs[4]); pre-generation budgets on the headline now that an
# episode is generated faster
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
export TMPDIR=/tmp
Q="--no-ppo --no-cpu-baseline --no-dropin --no-pmc-traffic --no-worst-case --no-other-configs"
pick() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "hh", d["roofline"]["launch_ms"], d["roofline"]["launch_ms_device"]["median"], d.get("step_decomposition", {}).get("median_us"), d.get("step_decomposition", {}).get("median_gap_us"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
timeout 900 python -m pytest tests/test_gpu_env.py tests/test_gpu_collect.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $O/pytest1.log
timeout 300 python bench.py --humans 50 --randomized --envs 8192 --steps 60 --warmup 10 --dephase 120 $Q > $O/c4.json 2> $O/c4.err; pick $O/c4.json
for B in 55 30 35 40 45 55; do
  timeout 200 python bench.py --steps 200 --warmup 30 $Q --pregen-budget-us $B > $O/pg_$B.json 2> $O/pg_$B.err; pick $O/pg_$B.json
done
