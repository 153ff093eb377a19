#!/bin/bash
# round 5, call B: the whole GPU suite on the merged build (placement loops 64 candidates per pass, plan builders, lp3 pairs, ADVICE fixes,
# the new training-size tests), then the driver's bench command with the other-config legs
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5b/bench_driver.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("headline", d["value"], d["ms_per_step"], "frac", r["frac"], r.get("frac_of_dense_bf16_algorithmic"), r.get("whole_step_frac"), r.get("l2_weight_stream_TBps"))
    print("decomp", r.get("step_decomposition_us"))
    print("ppo", d.get("ppo"))
    for c in d.get("other_baseline_configs_1gpu", []):
        print(c.get("config", "")[:40], c.get("env_steps_per_s"), c.get("ms_per_step"), c.get("kernel_median_us"), c.get("error"))
    print("cpu", d.get("cpu_baseline", {}).get("value"), "dropin", (d.get("dropin_train_loop") or {}).get("env_steps_per_s"))
except Exception as e:
    print("ERR", e); print(open("gpurun_out/r5b/bench_driver.err").read()[-2000:])
PY
