#!/bin/bash
# round 4, call A: full GPU suite + the driver's bench command + the long-window bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/a/pytest.log
tail -5 gpurun_out/a/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --timeline-out gpurun_out/a/timeline_20.txt > gpurun_out/a/bench_driver.json 2> gpurun_out/a/bench_driver.err; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-ppo --no-cpu-baseline --no-worst-case > gpurun_out/a/bench_driver2.json 2>> gpurun_out/a/bench_driver.err
timeout 300 python bench.py --no-ppo --no-cpu-baseline > gpurun_out/a/bench_long.json 2> gpurun_out/a/bench_long.err
python - <<'PY'
import json
for f in ("bench_driver", "bench_driver2", "bench_long"):
    try:
        d = json.loads(open("gpurun_out/a/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "ev", r["launch_ms_events"], "dev", r["launch_ms_device"], "rn", r["rn_fused_launch_ms_device"], "enq", d.get("host_enqueue_ms_per_step"))
        print("   decomp", d.get("step_decomposition"))
        if "ppo" in d: print("   ppo", d["ppo"].get("samples_per_s"), d["ppo"].get("update_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
