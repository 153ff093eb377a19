#!/bin/bash
# round 4, call G: the robot-node sequence as one forward / one backward call -- gradients vs the module path and the CPU graph, the reference
# PPO.update golden, then the PPO leg with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ppo.py -m gpu -x -q > gpurun_out/g/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/g/pytest.log
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin"
CN_TRAIN_FUSED_RN=0 $B > gpurun_out/g/ppo_modules.json 2> gpurun_out/g/err.log
$B > gpurun_out/g/ppo_fused.json 2>> gpurun_out/g/err.log
CN_TRAIN_FUSED_RN=0 $B > gpurun_out/g/ppo_modules2.json 2>> gpurun_out/g/err.log
$B > gpurun_out/g/ppo_fused2.json 2>> gpurun_out/g/err.log
python - <<'PY'
import json
for f in ("ppo_modules", "ppo_fused", "ppo_modules2", "ppo_fused2"):
    try:
        d = json.loads(open("gpurun_out/g/%s.json" % f).read().strip().splitlines()[-1])
        p = d["ppo"]
        print(f, p.get("samples_per_s"), "rollout_s", p.get("rollout_s"), "update_s", p.get("update_s"), "value_loss", p.get("value_loss"), p.get("error"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids gpurun_out/g/err.log | tail -5
