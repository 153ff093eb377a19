#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ppo.py -x -q -m gpu 2>&1 | tail -4
OUT=$GRAFT_REPO_ROOT/gpurun_out/t
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o u -- python $GRAFT_REPO_ROOT/examples/train_ppo.py --updates 4 > $OUT/update.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python examples/train_ppo.py --updates 4   (4 rollouts of 30 steps + 4 PPO updates = 40 optimiser steps, E=4096, H=20)" > $OUT/update_kernel_trace.txt 2>&1
grep "gru_seq" $OUT/update_kernel_trace.txt | cut -c1-175
