#!/bin/bash
# A/B in one box: 8 plan builders (quaternary level search, reciprocals) + pair-spreading lane kernel vs the shipped pair
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/d
timeout 900 python -m pytest tests/test_gpu_row_plan.py tests/test_gpu_env.py -m gpu -x -q -k "plan or nonrand or rand_r or golden" > gpurun_out/d/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/d/pytest.log
B="timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-ppo --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic"
OLD=$GRAFT_REPO_ROOT/.ab/libcrowdnav_hip_old.so
CN_HIP_LIB=$OLD $B > gpurun_out/d/old1.json 2> gpurun_out/d/err.log
$B > gpurun_out/d/new1.json 2>> gpurun_out/d/err.log
CN_HIP_LIB=$OLD $B > gpurun_out/d/old2.json 2>> gpurun_out/d/err.log
$B > gpurun_out/d/new2.json 2>> gpurun_out/d/err.log
python - <<'PY'
import json
for f in ("old1", "new1", "old2", "new2"):
    try:
        d = json.loads(open("gpurun_out/d/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-6s" % f, d["value"], d["ms_per_step"], "frac", r["frac"], "ev", r["launch_ms_events"]["median"], "dev", r["launch_ms_device"]["median"])
        print("      decomp", d["step_decomposition"]["median_us"], d["step_decomposition"]["median_gap_us"], d["step_decomposition"].get("median_step_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids gpurun_out/d/err.log | tail -5
