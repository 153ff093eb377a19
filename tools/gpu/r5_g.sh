#!/bin/bash
# round 5, call G: placement with the fp32 packed screen + the straddling candidate inside the next pass: bit-exact simulator suites, configs[4]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
Q="--no-ppo --no-cpu-baseline --no-dropin --no-pmc-traffic --no-worst-case --no-other-configs"
timeout 900 python -m pytest tests/test_gpu_env.py tests/test_gpu_collect.py tests/test_gpu_boundary.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $O/pytest1.log
timeout 300 python bench.py --humans 50 --randomized --envs 8192 --steps 60 --warmup 10 --dephase 120 $Q > $O/c4.json 2> $O/c4.err
timeout 200 python bench.py --steps 200 --warmup 30 $Q > $O/main200.json 2> $O/main200.err
python - <<'PY'
import json
for f in ("c4", "main200"):
    try:
        d = json.loads(open("gpurun_out/r5g/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("step_decomposition", {}).get("median_us"), d.get("step_decomposition", {}).get("median_gap_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
python tests/soak_gpu.py > $O/soak.log 2>&1 & SOAK=$!
sleep 170; kill $SOAK 2>/dev/null; tail -12 $O/soak.log
