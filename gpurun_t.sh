cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_all.log 2>&1; tail -4 gpurun_out/t_all.log
timeout 600 python bench.py --steps 300 --warmup 100 --no-cpu-baseline > gpurun_out/bench2.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench2.log
cd /tmp; rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 100 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "trace" > $GRAFT_REPO_ROOT/gpurun_out/r01_trace_now.txt 2>&1
