#!/bin/bash
# The round's artefacts (run through gpurun; outputs under gpurun_out/final_<tag>/, copy what should be judged into profiles/):
# kernel trace + four PMC passes + traffic JSON + update trace (tools/profile_step.sh), three PMC passes over one PPO update + digest
# (tools/gpu/update_profile.sh), the driver's bench command in full with the stamped timeline, the long-window bench, the 8-rank plumbing
# run, 60 training updates, kernel traces of the other three BASELINE config shapes.
#   gpurun --timeout 3000 -- 'bash tools/gpu/final_artifacts.sh r06'
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/profile_step.sh $TAG --with-update > $OUT/profile_step.log 2>&1; echo "profile_step rc=$?"
bash tools/gpu/update_profile.sh $TAG > $OUT/update_profile.log 2>&1; echo "update_profile rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --timeline-out $OUT/step_timeline.txt > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver rc=$?"
timeout 400 python bench.py --no-ppo --no-cpu-baseline --no-dropin --no-pmc-traffic --no-other-configs > $OUT/bench_long.json 2> $OUT/bench_long.err; echo "bench long rc=$?"
timeout 600 python bench.py --gpus 8 --same-gpu --dist-backend gloo --envs 512 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs > $OUT/bench_8rank_gloo.json 2> $OUT/bench_8rank.err; echo "bench8 rc=$?"
timeout 300 python examples/train_ppo.py --updates 60 > $OUT/train_60_updates.jsonl 2> $OUT/train_60.err; echo "train60 rc=$?"; tail -1 $OUT/train_60_updates.jsonl | cut -c1-300
C="--no-ppo --no-cpu-baseline --no-dropin --no-pmc-traffic --no-worst-case --no-other-configs"
i=2
for A in "--env-name CrowdSimPred-v0 --envs 4096 --steps 60 --warmup 20" "--env-name CrowdSimPredRealGST-v0 --envs 2048 --steps 60 --warmup 20" "--humans 50 --randomized --envs 8192 --steps 40 --warmup 10 --dephase 120"; do
  cd /tmp; rm -rf /tmp/profc
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/profc -o t -- python $GRAFT_REPO_ROOT/bench.py $A $C > $GRAFT_REPO_ROOT/$OUT/configs${i}.log 2>&1
  python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/profc -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py $A $C" > $GRAFT_REPO_ROOT/$OUT/configs${i}_trace.txt 2>&1
  cd $GRAFT_REPO_ROOT; i=$((i+1))
done
python - <<PY
import json
for f in ("bench_driver", "bench_long", "bench_8rank_gloo"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "ppo", (d.get("ppo") or {}).get("samples_per_s"), (d.get("ppo") or {}).get("update_s"))
        for x in d.get("other_baseline_configs_1gpu") or []:
            print("   ", x.get("config", "")[:40], x.get("env_steps_per_s"), x.get("ms_per_step"), x.get("error"))
    except Exception as e:
        print(f, "ERR", e)
PY
