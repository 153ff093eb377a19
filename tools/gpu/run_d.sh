#!/bin/bash
# round 4, call D: 8 plan builders + pair-spreading lane kernel -- plan invariants, bit-exact env suite, A/B against the previous build in one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/d
timeout 900 python -m pytest tests/test_gpu_row_plan.py tests/test_gpu_env.py tests/test_gpu_fullsize.py tests/test_gpu_eval.py -m gpu -x -q > gpurun_out/d/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/d/pytest.log
B="timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-ppo --no-cpu-baseline --no-worst-case"
OLD=$GRAFT_REPO_ROOT/.ab/libcrowdnav_hip_old.so
CN_HIP_LIB=$OLD $B > gpurun_out/d/old1.json 2> gpurun_out/d/err.log
$B --timeline-out gpurun_out/d/timeline_new.txt > gpurun_out/d/new1.json 2>> gpurun_out/d/err.log
CN_HIP_LIB=$OLD $B > gpurun_out/d/old2.json 2>> gpurun_out/d/err.log
$B > gpurun_out/d/new2.json 2>> gpurun_out/d/err.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-ppo --no-cpu-baseline --no-worst-case > gpurun_out/d/new_driver.json 2>> gpurun_out/d/err.log
python - <<'PY'
import json
for f in ("old1", "new1", "old2", "new2", "new_driver"):
    try:
        d = json.loads(open("gpurun_out/d/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-11s" % f, d["value"], d["ms_per_step"], "frac", r["frac"], "ev", r["launch_ms_events"]["median"], "dev", r["launch_ms_device"]["median"], "rn", r["rn_fused_launch_ms_device"]["median"])
        print("      decomp", d["step_decomposition"]["median_us"], d["step_decomposition"]["median_gap_us"], d["step_decomposition"].get("median_step_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids gpurun_out/d/err.log | tail -5
