#!/bin/bash
# PMC passes over one PPO update (examples/train_ppo.py --updates 1): what bounds gemm3p_tn / gemm3p_nt / hh_fused<true> / attention backward
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/j
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcout
  timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcout -o pmc -- python $GRAFT_REPO_ROOT/examples/train_ppo.py --updates 1 > $OUT/pmc_$i.log 2>&1
  python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/pmcout -name "*.db" | head -1) "pmc pass $i: rocprofv3 --kernel-trace --pmc $C -- python examples/train_ppo.py --updates 1" > $OUT/pmc_$i.txt 2>&1
  tail -2 $OUT/pmc_$i.log
done
