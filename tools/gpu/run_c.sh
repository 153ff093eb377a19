#!/bin/bash
# round 4, call C: deferred tail -- bit-exactness test + A/B of the step inside one box (inline vs deferred, pregen budgets)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "deferred or stale" > gpurun_out/c/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c/pytest.log
B="timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-ppo --no-cpu-baseline --no-worst-case"
$B --tail inline > gpurun_out/c/inline.json 2> gpurun_out/c/err.log
$B --tail deferred --timeline-out gpurun_out/c/timeline_deferred.txt > gpurun_out/c/deferred.json 2>> gpurun_out/c/err.log
$B --tail deferred --pregen-budget-us 35 > gpurun_out/c/deferred_p35.json 2>> gpurun_out/c/err.log
$B --tail deferred --pregen-budget-us 150 > gpurun_out/c/deferred_p150.json 2>> gpurun_out/c/err.log
$B --tail inline > gpurun_out/c/inline2.json 2>> gpurun_out/c/err.log
$B --tail deferred > gpurun_out/c/deferred2.json 2>> gpurun_out/c/err.log
python - <<'PY'
import json
for f in ("inline", "deferred", "deferred_p35", "deferred_p150", "inline2", "deferred2"):
    try:
        d = json.loads(open("gpurun_out/c/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-14s" % f, d["value"], d["ms_per_step"], "frac", r["frac"], "ev", r["launch_ms_events"]["median"], "dev", r["launch_ms_device"]["median"], "rn", r["rn_fused_launch_ms_device"]["median"])
        print("      decomp", d["step_decomposition"]["median_us"], d["step_decomposition"]["median_gap_us"], d["step_decomposition"].get("median_step_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 gpurun_out/c/err.log
