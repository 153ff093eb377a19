#!/bin/bash
# Profile of the PPO update on the GPU box (run through gpurun): kernel trace of 4 rollouts + 4 updates, three PMC passes over one
# update, digest.  Output: gpurun_out/upd_<tag>/ -- copy what should be judged into profiles/.
#   gpurun --timeout 1500 -- 'bash tools/gpu/update_profile.sh r06a [--no-pmc]'
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/upd_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o u -- python $GRAFT_REPO_ROOT/examples/train_ppo.py --updates 4 > $OUT/update.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python examples/train_ppo.py --updates 4   (4 rollouts of 30 steps + 4 PPO updates = 40 optimiser steps, E=4096, H=20)" > $OUT/update_kernel_trace.txt 2>&1
if [ "$2" != "--no-pmc" ]; then
i=1
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcout
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcout -o pmc -- python $GRAFT_REPO_ROOT/examples/train_ppo.py --updates 1 > $OUT/pmc_$i.log 2>&1
  python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/pmcout -name "*.db" | head -1) "pmc pass $i: rocprofv3 --kernel-trace --pmc $C -- python examples/train_ppo.py --updates 1" > $OUT/pmc_$i.txt 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_digest.py $OUT/pmc_2.txt $OUT/pmc_3.txt $OUT/pmc_4.txt > $OUT/update_pmc.txt 2>&1
fi
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --no-pmc-traffic --no-other-configs --no-worst-case > $OUT/bench_ppo.json 2> $OUT/bench_ppo.err
head -30 $OUT/update_kernel_trace.txt | cut -c1-200
tail -c 1500 $OUT/bench_ppo.json
