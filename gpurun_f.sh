cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 400 --warmup 100 > gpurun_out/bench_bf16x3.log 2>&1; echo "rc=$?" >> gpurun_out/bench_bf16x3.log
timeout 600 python bench.py --steps 300 --warmup 100 --gemm fp32 --no-cpu-baseline > gpurun_out/bench_fp32.log 2>&1; echo "rc=$?" >> gpurun_out/bench_fp32.log
cd /tmp; rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 100 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "trace" > $GRAFT_REPO_ROOT/gpurun_out/r01_final_trace.txt 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1)); rm -rf /tmp/pmcout
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcout -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 30 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$i.log 2>&1
  python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/pmcout -name "*.db" | head -1) "pmc pass $i: $C" > $GRAFT_REPO_ROOT/gpurun_out/r01_final_pmc_$i.txt 2>&1
done
