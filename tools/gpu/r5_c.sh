#!/bin/bash
# round 5, call C: placement walk on squares + annulus filter (bit-exact simulator tests), policy variants on the device, training-size tests,
# configs[4] rate with its kernel trace
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_env.py tests/test_gpu_policy_variants.py tests/test_gpu_train_scale.py tests/test_gpu_policy.py tests/test_gpu_train.py tests/test_gpu_collect.py -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-220
Q="--no-ppo --no-cpu-baseline --no-dropin --no-pmc-traffic --no-worst-case --no-other-configs"
cd /tmp; rm -rf /tmp/p4
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/bench.py --humans 50 --randomized --envs 8192 --steps 40 --warmup 10 --dephase 120 $Q > $O/c4.json 2> $O/c4.err
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/p4 -name "*.db" | head -1) "configs[4] shape: 50 randomised humans, 8192 envs" > $O/c4_trace.txt 2>&1
head -9 $O/c4_trace.txt | cut -c1-150
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c/c4.json").read().strip().splitlines()[-1])
print("c4", d["value"], d["ms_per_step"], d.get("step_decomposition", {}).get("median_us"))
PY
