#!/bin/bash
# ORCA tail with a bounded grid (workgroups per CU): does the robot-node kernel get its CUs sooner?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ae
B="timeout 300 python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-ppo"
for cap in 0 6 5 4 3 0 6; do
  CN_LP3_WGS_PER_CU=$cap timeout 200 $B --timeline-out gpurun_out/ae/tl_$cap.txt 2>> gpurun_out/ae/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lp3 wgs/CU $cap', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launch_ms'))"
done
python - <<'PY'
import numpy as np
for v in ("0","6","4"):
    rows=[l.split() for l in open('gpurun_out/ae/tl_%s.txt'%v) if not l.startswith('#')]
    st={}
    for s,k,a,b,d in rows: st.setdefault(int(s),{})[k]=(float(a),float(b))
    need=('env_step','orca_lane','row_plan','hh_fused','rn_fused','env_pregen','orca_lp3')
    S=[s for s in sorted(st)[1:-1] if all(k in st[s] for k in need) and s+1 in st and 'env_step' in st[s+1]]
    def avg(f): return np.mean([f(st[s],st[s+1]) for s in S])
    print(v, "hh %.1f | hh->rn %.1f | rn %.1f | rn->env %.1f | lp3 start-hh end %.1f dur %.1f, end - rn end %.1f | step %.1f"%(
        avg(lambda x,y:x['hh_fused'][1]-x['hh_fused'][0]),
        avg(lambda x,y:x['rn_fused'][0]-x['hh_fused'][1]), avg(lambda x,y:x['rn_fused'][1]-x['rn_fused'][0]), avg(lambda x,y:y['env_step'][0]-x['rn_fused'][1]),
        avg(lambda x,y:x['orca_lp3'][0]-x['hh_fused'][1]), avg(lambda x,y:x['orca_lp3'][1]-x['orca_lp3'][0]), avg(lambda x,y:x['orca_lp3'][1]-x['rn_fused'][1]),
        avg(lambda x,y:y['env_step'][0]-x['env_step'][0])))
PY
grep -v amdgpu.ids gpurun_out/ae/err.log | tail -3
