#!/bin/bash
# round 5, call A: (1) A/B of the parked step candidates on the headline (CN_HIP_LIB builds under .ab/), (2) the other BASELINE configs
# on one GPU with kernel traces, (3) the simulator's bit-exact tests on the merged candidate build
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
Q="--no-ppo --no-cpu-baseline --no-dropin --no-pmc-traffic --no-worst-case"
pick() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "hh", d["roofline"]["launch_ms"], d.get("step_decomposition", {}).get("median_us"), d.get("step_decomposition", {}).get("median_gap_us"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for pass in 1 2; do
  for L in main B C D; do
    if [ $L == main ]; then unset CN_HIP_LIB; else export CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$L.so; fi
    timeout 200 python bench.py --steps 200 --warmup 30 $Q > $O/ab_${L}_$pass.json 2> $O/ab_${L}_$pass.err
    pick $O/ab_${L}_$pass.json
  done
done
unset CN_HIP_LIB
# configs[2], [3], [4] shapes on one GPU, main build
cd /tmp
rm -rf /tmp/p2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o t -- python $GRAFT_REPO_ROOT/bench.py --env-name CrowdSimPred-v0 --steps 100 --warmup 30 $Q > $O/c2.json 2> $O/c2.err
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/p2 -name "*.db" | head -1) "configs[2] shape: CrowdSimPred-v0 4096 envs" > $O/c2_trace.txt 2>&1
pick $O/c2.json
rm -rf /tmp/p3; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p3 -o t -- python $GRAFT_REPO_ROOT/bench.py --env-name CrowdSimPredRealGST-v0 --envs 2048 --steps 100 --warmup 30 $Q > $O/c3.json 2> $O/c3.err
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/p3 -name "*.db" | head -1) "configs[3] shape: CrowdSimPredRealGST-v0 2048 envs, GST in the loop" > $O/c3_trace.txt 2>&1
pick $O/c3.json
rm -rf /tmp/p4; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/bench.py --humans 50 --randomized --envs 8192 --steps 10 --warmup 4 --dephase 40 $Q > $O/c4.json 2> $O/c4.err
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/p4 -name "*.db" | head -1) "configs[4] shape: 50 randomised humans, 8192 envs" > $O/c4_trace.txt 2>&1
pick $O/c4.json
cd $GRAFT_REPO_ROOT
head -14 $O/c3_trace.txt | cut -c1-150
head -8 $O/c4_trace.txt | cut -c1-150
CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libD.so timeout 600 python -m pytest tests/test_gpu_env.py tests/test_gpu_row_plan.py tests/test_gpu_eval.py -x -q > $O/pytest_D.log 2>&1; echo "pytest D rc=$?"; tail -3 $O/pytest_D.log
