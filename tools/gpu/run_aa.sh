#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/aa
timeout 900 python -m pytest tests/test_gpu_env.py tests/test_gpu_row_plan.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
B="timeout 300 python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-ppo"
for cfg in "off 55" "on 55" "off 55" "on 55" "on 40" "on 70"; do
  set -- $cfg
  timeout 200 $B --forward-gate $1 --pregen-budget-us $2 --timeline-out gpurun_out/aa/tl_$1_$2.txt 2>> gpurun_out/aa/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gate $1 budget $2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launch_ms'))"
done
python - <<'PY'
import numpy as np
for v in ("off_55","on_55"):
    rows=[l.split() for l in open('gpurun_out/aa/tl_%s.txt'%v) if not l.startswith('#')]
    st={}
    for s,k,a,b,d in rows: st.setdefault(int(s),{})[k]=(float(a),float(b))
    need=('env_step','orca_lane','row_plan','hh_fused','rn_fused','env_pregen','orca_lp3')
    S=[s for s in sorted(st)[1:-1] if all(k in st[s] for k in need) and s+1 in st and 'env_step' in st[s+1]]
    def avg(f): return np.mean([f(st[s],st[s+1]) for s in S])
    print(v, "env %.1f | ->lane %.1f | lane %.1f plan %.1f | ->hh %.1f | hh %.1f | hh->rn %.1f | rn %.1f | rn->env %.1f | lp3 start-hh start %.1f dur %.1f, end - rn end %.1f | pregen start - env end %.1f dur %.1f | step %.1f"%(
        avg(lambda x,y:x['env_step'][1]-x['env_step'][0]), avg(lambda x,y:x['orca_lane'][0]-x['env_step'][1]), avg(lambda x,y:x['orca_lane'][1]-x['orca_lane'][0]),
        avg(lambda x,y:x['row_plan'][1]-x['row_plan'][0]),
        avg(lambda x,y:x['hh_fused'][0]-max(x['row_plan'][1],x['orca_lane'][1])), avg(lambda x,y:x['hh_fused'][1]-x['hh_fused'][0]),
        avg(lambda x,y:x['rn_fused'][0]-x['hh_fused'][1]), avg(lambda x,y:x['rn_fused'][1]-x['rn_fused'][0]), avg(lambda x,y:y['env_step'][0]-x['rn_fused'][1]),
        avg(lambda x,y:x['orca_lp3'][0]-x['hh_fused'][0]), avg(lambda x,y:x['orca_lp3'][1]-x['orca_lp3'][0]), avg(lambda x,y:x['orca_lp3'][1]-x['rn_fused'][1]),
        avg(lambda x,y:x['env_pregen'][0]-x['env_step'][1]), avg(lambda x,y:x['env_pregen'][1]-x['env_pregen'][0]), avg(lambda x,y:y['env_step'][0]-x['env_step'][0])))
PY
grep -v amdgpu.ids gpurun_out/aa/err.log | tail -5
