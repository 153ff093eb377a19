#!/bin/bash
# A/B of build flags of hh_fused.hip on the PPO leg (run through gpurun): the default build, then each quoted HHFLAGS string.
#   gpurun -- 'bash tools/gpu/hh_ab.sh "-DHH_TRAIN_STORE_AUX=0"'
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --no-pmc-traffic --no-other-configs --no-worst-case"
run() {
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'step_ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'update_s', d['ppo']['update_s'], 'samples/s', d['ppo']['samples_per_s'])"
}
run default; run default
for F in "$@"; do
  touch crowdnav_prediction_attngraph_amd/csrc/hh_fused.hip
  make -s -C crowdnav_prediction_attngraph_amd/csrc HHFLAGS="$F" 2>&1 | grep -E "error" | head -5
  run "[$F]"; run "[$F]"
done
