cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_all.log 2>&1; tail -12 gpurun_out/t_all.log
timeout 600 python examples/train_ppo.py --num-processes 4096 --updates 3 > gpurun_out/train1.log 2>&1; echo "train rc=$?" >> gpurun_out/train1.log
