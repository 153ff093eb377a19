#!/bin/bash
# Profiles of the rollout step on the GPU box (run through gpurun): kernel trace + four PMC passes + the traffic JSON, written to
# gpurun_out/prof_<tag>/ -- copy the summaries you want judged into profiles/.
#   gpurun -- 'bash tools/profile_step.sh r03 [--with-update]'
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 100 --warmup 60 --no-cpu-baseline --no-ppo --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs"
cd /tmp; rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/profiles/summarize.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py $ARGS   (240 de-phase + 60 warm-up + 100 timed steps)" > $OUT/kernel_trace.txt 2>&1
python $GRAFT_REPO_ROOT/tools/step_timeline.py $DB > $OUT/timeline.txt 2>&1
PARGS="--steps 20 --warmup 20 --no-cpu-baseline --no-ppo --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pmcout
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcout -o pmc -- python $GRAFT_REPO_ROOT/bench.py $PARGS > $OUT/pmc_$i.log 2>&1
  python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/pmcout -name "*.db" | head -1) "pmc pass $i: rocprofv3 --kernel-trace --pmc $C -- python bench.py $PARGS" > $OUT/pmc_$i.txt 2>&1
done
python $GRAFT_REPO_ROOT/tools/mk_traffic.py $OUT $OUT/pmc_traffic.json
if [ "$2" == "--with-update" ]; then
  rm -rf /tmp/prof
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o u -- python $GRAFT_REPO_ROOT/examples/train_ppo.py --updates 4 > $OUT/update.log 2>&1
  python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python examples/train_ppo.py --updates 4   (4 rollouts of 30 steps + 4 PPO updates = 40 optimiser steps, E=4096, H=20)" > $OUT/update_kernel_trace.txt 2>&1
fi
cd $GRAFT_REPO_ROOT
head -12 $OUT/kernel_trace.txt | cut -c1-180
cat $OUT/timeline.txt
