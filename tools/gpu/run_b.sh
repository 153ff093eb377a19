#!/bin/bash
# round 4, call B: new tests (stride, GST history, evaluation default mode, lazy infos) + bench with the light stamps and the sim->forward order
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b
timeout 900 python -m pytest tests/test_gst_host.py tests/test_gpu_eval.py tests/test_gpu_train.py tests/test_gpu_boundary.py tests/test_gpu_collect.py -m gpu -x -q > gpurun_out/b/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 gpurun_out/b/pytest1.log
timeout 900 python -m pytest tests/test_gpu_env.py -m gpu -x -q -k "stride or golden" > gpurun_out/b/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -3 gpurun_out/b/pytest2.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-ppo --no-cpu-baseline --timeline-out gpurun_out/b/timeline_20.txt > gpurun_out/b/bench_driver.json 2> gpurun_out/b/bench_driver.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-ppo --no-cpu-baseline --no-worst-case > gpurun_out/b/bench_driver2.json 2>> gpurun_out/b/bench_driver.err
timeout 300 python bench.py --no-ppo --no-cpu-baseline --no-worst-case > gpurun_out/b/bench_long.json 2> gpurun_out/b/bench_long.err
python - <<'PY'
import json
for f in ("bench_driver", "bench_driver2", "bench_long"):
    try:
        d = json.loads(open("gpurun_out/b/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "ev", {k: r["launch_ms_events"][k] for k in ("median", "min", "max", "samples")}, "dev", {k: r["launch_ms_device"][k] for k in ("median", "min", "max")}, "rn", r["rn_fused_launch_ms_device"]["median"], "enq", d.get("host_enqueue_ms_per_step"))
        print("   intervals", d.get("device_step_interval_us"))
        print("   decomp", d.get("step_decomposition"))
    except Exception as e:
        print(f, "ERR", e)
PY
