cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_all.log 2>&1; tail -4 gpurun_out/t_all.log
timeout 600 python bench.py --steps 300 --warmup 100 --no-cpu-baseline > gpurun_out/bench2.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench2.log
timeout 600 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --env-name CrowdSimPredRealGST-v0 --envs 2048 > gpurun_out/bench_gst.log 2>&1; echo "rc=$?" >> gpurun_out/bench_gst.log
