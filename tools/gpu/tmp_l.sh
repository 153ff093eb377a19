#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5l; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o u -- python $GRAFT_REPO_ROOT/examples/train_ppo.py --updates 4 > $O/update.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python examples/train_ppo.py --updates 4   (4 rollouts of 30 steps + 4 PPO updates = 40 optimiser steps, E=4096, H=20)" > $O/update_kernel_trace.txt 2>&1
cd $GRAFT_REPO_ROOT
grep "hr_attention" $O/update_kernel_trace.txt | cut -c1-150
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --timeline-out $O/step_timeline.txt > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5l/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], "frac", r["frac"], r["launch_ms"], r["launch_ms_device"]["median"], r["launch_ms_device"]["frac_at_median"], r.get("frac_of_dense_bf16_algorithmic"), r.get("whole_step_frac"), r.get("l2_weight_stream_TBps"), r["traffic"])
print("decomp", r.get("step_decomposition_us"))
print("ppo", d["ppo"]["samples_per_s"], d["ppo"]["update_s"], d["ppo"]["rollout_s"], d["ppo"]["roofline"]["frac"])
print("worst", d["worst_case_all_detected"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"], d["gpu_over_cpu"], d["speedup_vs_reference"]["vs_reference_python_scaled_to_this_box"])
for c in d.get("other_baseline_configs_1gpu", []):
    print(c.get("config", "")[:40], c.get("env_steps_per_s"), c.get("ms_per_step"), c.get("error"))
print("dropin", d["dropin_train_loop"]["env_steps_per_s"], "interval", d["device_step_interval_us"]["median"])
PY
