cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_all.log 2>&1; tail -3 gpurun_out/t_all.log
timeout 600 python bench.py --steps 300 --warmup 100 --no-cpu-baseline > gpurun_out/bench2.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench2.log
timeout 600 python examples/train_ppo.py --num-processes 4096 --updates 3 > gpurun_out/train1.log 2>&1; echo "train rc=$?" >> gpurun_out/train1.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01b -o r01b -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 100 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
