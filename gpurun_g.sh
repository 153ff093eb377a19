cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 30 --warmup 10 --dist-backend gloo --same-gpu --envs 1024 > gpurun_out/bench_2rank.log 2>&1; echo "rc=$?" >> gpurun_out/bench_2rank.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_1rank_torchrun.log 2>&1; echo "rc=$?" >> gpurun_out/bench_1rank_torchrun.log
