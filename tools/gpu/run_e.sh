#!/bin/bash
# round 4, call E: the pre-generation riding in the lane kernel's launch -- bit-exactness (env suite incl. budget independence) + A/B in one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e
timeout 900 python -m pytest tests/test_gpu_env.py -m gpu -x -q -k "budget or pregen or varnum_h20_nonrand or varnum_h5_rand or golden" > gpurun_out/e/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/e/pytest.log
B="timeout 200 python bench.py --gpus 1 --steps 100 --warmup 10 --no-ppo --no-cpu-baseline --no-worst-case"
CN_PREGEN_INLANE=0 $B > gpurun_out/e/side1.json 2> gpurun_out/e/err.log
$B --timeline-out gpurun_out/e/timeline_ride.txt > gpurun_out/e/ride40_1.json 2>> gpurun_out/e/err.log
CN_PREGEN_RIDE_CAP_US=30 $B > gpurun_out/e/ride30.json 2>> gpurun_out/e/err.log
CN_PREGEN_RIDE_CAP_US=48 $B > gpurun_out/e/ride48.json 2>> gpurun_out/e/err.log
CN_PREGEN_INLANE=0 $B > gpurun_out/e/side2.json 2>> gpurun_out/e/err.log
$B > gpurun_out/e/ride40_2.json 2>> gpurun_out/e/err.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-ppo --no-cpu-baseline --no-worst-case > gpurun_out/e/ride_driver.json 2>> gpurun_out/e/err.log
python - <<'PY'
import json
for f in ("side1", "ride40_1", "ride30", "ride48", "side2", "ride40_2", "ride_driver"):
    try:
        d = json.loads(open("gpurun_out/e/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-11s" % f, d["value"], d["ms_per_step"], "frac", r["frac"], "ev", r["launch_ms_events"]["median"], "dev", r["launch_ms_device"]["median"], "rn", r["rn_fused_launch_ms_device"]["median"])
        print("      decomp", d["step_decomposition"]["median_us"], d["step_decomposition"]["median_gap_us"], d["step_decomposition"].get("median_step_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids gpurun_out/e/err.log | tail -5
