#!/bin/bash
# round 5, call D: headline A/B -- env_step at 4 waves per SIMD (libE), pre-generation budgets, stamped timeline of hh_fused's workgroup starts
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
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
for pass in 1 2; do
  unset CN_HIP_LIB
  timeout 200 python bench.py --steps 200 --warmup 30 $Q --timeline-out $O/timeline_main_$pass.txt > $O/ab_main_$pass.json 2> $O/ab_main_$pass.err; pick $O/ab_main_$pass.json
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libE.so timeout 200 python bench.py --steps 200 --warmup 30 $Q > $O/ab_E_$pass.json 2> $O/ab_E_$pass.err; pick $O/ab_E_$pass.json
done
for B in 25 35 45 70; do
  timeout 200 python bench.py --steps 200 --warmup 30 $Q --pregen-budget-us $B > $O/pg_$B.json 2> $O/pg_$B.err; pick $O/pg_$B.json
done
grep "^#   step" $O/timeline_main_1.txt | head -12
timeout 600 python -m pytest tests/test_gpu_train_scale.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
