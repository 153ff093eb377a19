#!/bin/bash
# round 5, call F: pre-generation in the tail launch with ONE generator wavefront per block (2.5 KB of static LDS per block instead of 10):
# A/B against the round-4 placement (CN_PREGEN_SEPARATE=1) and the tail at 6 wavefronts per SIMD without scratch (libF6)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
Q="--no-ppo --no-cpu-baseline --no-dropin --no-pmc-traffic --no-worst-case --no-other-configs"
pick() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "hh", d["roofline"]["launch_ms"], d["roofline"]["launch_ms_device"]["median"], d.get("step_decomposition", {}).get("median_us"), d.get("step_decomposition", {}).get("median_gap_us"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for pass in 1 2; do
  unset CN_HIP_LIB CN_PREGEN_SEPARATE
  timeout 200 python bench.py --steps 200 --warmup 30 $Q --timeline-out $O/timeline_main_$pass.txt > $O/ab_main_$pass.json 2> $O/ab_main_$pass.err; pick $O/ab_main_$pass.json
  CN_PREGEN_SEPARATE=1 timeout 200 python bench.py --steps 200 --warmup 30 $Q > $O/ab_sep_$pass.json 2> $O/ab_sep_$pass.err; pick $O/ab_sep_$pass.json
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libF6.so timeout 200 python bench.py --steps 200 --warmup 30 $Q > $O/ab_F6_$pass.json 2> $O/ab_F6_$pass.err; pick $O/ab_F6_$pass.json
done
grep "^#   step" $O/timeline_main_1.txt | head -4
timeout 200 python bench.py --steps 20 --warmup 5 $Q > $O/drv.json 2> $O/drv.err; pick $O/drv.json
timeout 600 python -m pytest tests/test_gpu_env.py tests/test_gpu_gst_train.py -m gpu -x -q > $O/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -3 $O/pytest1.log
