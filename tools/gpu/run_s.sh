#!/bin/bash
# whole GPU suite + the driver's bench command on the current build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s/bench_driver.json 2> gpurun_out/s/bench_driver.err; echo "bench driver rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "ppo", d["ppo"])
PY
