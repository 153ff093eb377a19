cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 -L 2>&1 | grep -iE "^\s*(Name|Counter)|FETCH|WRITE_SIZE|TCC_HIT|TCC_MISS|MFMA|SQ_WAIT|TA_BUSY|SQ_BUSY|LDS" | head -150 > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmcout
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcout -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 30 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$i.log 2>&1
  echo "pass $i rc=$?"
  python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/pmcout -name "*.db" | head -1) "pmc pass $i: $C" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$i.txt 2>&1
done
