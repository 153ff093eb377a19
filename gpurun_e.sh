cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 300 --warmup 100 --no-cpu-baseline > gpurun_out/bench2.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench2.log
