cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --env-name CrowdSimPredRealGST-v0 --envs 2048 > gpurun_out/bench_gst.log 2>&1; echo "rc=$?" >> gpurun_out/bench_gst.log
timeout 600 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --env-name CrowdSimPred-v0 > gpurun_out/bench_pred.log 2>&1; echo "rc=$?" >> gpurun_out/bench_pred.log
cd /tmp; rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 50 --no-cpu-baseline --env-name CrowdSimPredRealGST-v0 --envs 2048 > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "trace gst" > $GRAFT_REPO_ROOT/gpurun_out/r01_gst_trace.txt 2>&1
