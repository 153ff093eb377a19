#!/bin/bash
# round 4, call C3: any-order launches on ONE stream (CN_TAIL_MODE=2) vs inline vs deferred
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
CN_TAIL_MODE=2 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "deferred" > gpurun_out/c3/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c3/pytest.log
B="timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-ppo --no-cpu-baseline --no-worst-case"
$B --tail inline > gpurun_out/c3/inline.json 2> gpurun_out/c3/err.log
CN_TAIL_MODE=2 $B --tail deferred --pregen-budget-us 40 --timeline-out gpurun_out/c3/timeline_anyorder.txt > gpurun_out/c3/anyorder_p40.json 2>> gpurun_out/c3/err.log
CN_TAIL_MODE=2 $B --tail deferred --pregen-budget-us 30 > gpurun_out/c3/anyorder_p30.json 2>> gpurun_out/c3/err.log
CN_TAIL_MODE=2 $B --tail deferred > gpurun_out/c3/anyorder_p55.json 2>> gpurun_out/c3/err.log
$B --tail inline > gpurun_out/c3/inline2.json 2>> gpurun_out/c3/err.log
python - <<'PY'
import json
for f in ("inline", "anyorder_p40", "anyorder_p30", "anyorder_p55", "inline2"):
    try:
        d = json.loads(open("gpurun_out/c3/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-14s" % f, d["value"], d["ms_per_step"], "frac", r["frac"], "ev", r["launch_ms_events"]["median"], "dev", r["launch_ms_device"]["median"], "rn", r["rn_fused_launch_ms_device"]["median"])
        print("      decomp", d["step_decomposition"]["median_us"], d["step_decomposition"]["median_gap_us"], d["step_decomposition"].get("median_step_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids gpurun_out/c3/err.log | tail -5
head -24 gpurun_out/c3/timeline_anyorder.txt
