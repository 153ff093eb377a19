cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o upd -- python $GRAFT_REPO_ROOT/examples/train_ppo.py --num-processes 4096 --updates 4 > $GRAFT_REPO_ROOT/gpurun_out/prof_upd.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof -name "*.db" | head -1) "trace update" > $GRAFT_REPO_ROOT/gpurun_out/r01_update_trace.txt 2>&1
