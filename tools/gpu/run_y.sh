#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/y


B="timeout 300 python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-ppo"
for v in H F H F; do
  CN_HIP_LIB=$GRAFT_REPO_ROOT/.ab/lib$v.so $B --timeline-out gpurun_out/y/tl_$v.txt 2>> gpurun_out/y/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); dm=d.get('decomposition',{}).get('median_us',{}); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launch_ms'), {k:dm.get(k) for k in ('orca_lane','row_plan','hh_fused')})"
done
grep -v amdgpu.ids gpurun_out/y/err.log | tail -3
python - <<'PY'
import numpy as np
for v in "HF":
    rows=[l.split() for l in open('gpurun_out/y/tl_%s.txt'%v) if not l.startswith('#')]
    st={}
    for s,k,a,b,d in rows: st.setdefault(int(s),{})[k]=(float(a),float(b))
    need=('env_step','orca_lane','row_plan','hh_fused','rn_fused','env_pregen','orca_lp3')
    def avg(f): return np.mean([f(st[s]) for s in sorted(st)[1:-1] if all(k in st[s] for k in need)])
    print(v, "env %.1f | env->lane %.1f | lane %.1f plan %.1f (plan end - lane end %.1f) | ->hh %.1f | hh %.1f | hh->rn %.1f | rn %.1f | pregen end - hh start %.1f"%(
        avg(lambda x:x['env_step'][1]-x['env_step'][0]), avg(lambda x:x['orca_lane'][0]-x['env_step'][1]), avg(lambda x:x['orca_lane'][1]-x['orca_lane'][0]),
        avg(lambda x:x['row_plan'][1]-x['row_plan'][0]), avg(lambda x:x['row_plan'][1]-x['orca_lane'][1]),
        avg(lambda x:x['hh_fused'][0]-max(x['row_plan'][1],x['orca_lane'][1])), avg(lambda x:x['hh_fused'][1]-x['hh_fused'][0]),
        avg(lambda x:x['rn_fused'][0]-x['hh_fused'][1]), avg(lambda x:x['rn_fused'][1]-x['rn_fused'][0]), avg(lambda x:x['env_pregen'][1]-x['hh_fused'][0])))
PY
