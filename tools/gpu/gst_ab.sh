#!/bin/bash
# A/B of the GST kernels' tile sizes / workgroups per CU on configs[3] (run through gpurun): default build, then the variants given as
# quoted GSTFLAGS strings.   gpurun -- 'bash tools/gpu/gst_ab.sh "-DCN_GL_RT=3 -DCN_GST_WGS=2 -DCN_LS_ROWS=32" ...'
cd $GRAFT_REPO_ROOT
B="python bench.py --env-name CrowdSimPredRealGST-v0 --envs 2048 --steps 60 --warmup 20 --no-cpu-baseline --no-ppo --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs"
run() {
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('step_decomposition',{}).get('median_us'))"
}
run default
for F in "$@"; do
  touch crowdnav_prediction_attngraph_amd/csrc/gst.hip
  make -s -C crowdnav_prediction_attngraph_amd/csrc GSTFLAGS="$F" 2>&1 | grep -E "error" | head -5
  run "[$F]"
  python -m pytest tests/test_gst_host.py -x -q -m gpu 2>&1 | tail -1
done
