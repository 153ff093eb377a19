cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_all.log 2>&1; tail -3 gpurun_out/t_all.log
timeout 600 python bench.py --steps 300 --warmup 100 --no-cpu-baseline > gpurun_out/bench2.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench2.log
cd /tmp; rm -rf /tmp/pmcout
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcout -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 30 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_1.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/pmcout -name "*.db" | head -1) "pmc FETCH_SIZE" > $GRAFT_REPO_ROOT/gpurun_out/pmc_1b.txt 2>&1
