#!/bin/bash
# round 4, call F: episode-stats kernel test, the driver's bench command in full (PPO leg, CPU baseline, drop-in leg), the 8-rank plumbing run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_boundary.py -m gpu -x -q > gpurun_out/f/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/f/pytest.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/f/bench_driver.json 2> gpurun_out/f/err.log; echo "bench rc=$?"
timeout 600 python bench.py --gpus 8 --same-gpu --dist-backend gloo --envs 512 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case > gpurun_out/f/bench_8rank_gloo.json 2> gpurun_out/f/err8.log; echo "bench8 rc=$?"
python - <<'PY'
import json
for f in ("bench_driver", "bench_8rank_gloo"):
    try:
        d = json.loads(open("gpurun_out/f/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "ppo", d.get("ppo"))
        print("   dropin", d.get("dropin_train_loop")); print("   per_rank", d.get("per_rank_env_steps_per_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
grep -v amdgpu.ids gpurun_out/f/err.log | tail -5; grep -v amdgpu.ids gpurun_out/f/err8.log | tail -8
