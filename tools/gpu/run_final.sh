#!/bin/bash
# round 5 (same recipe as round 4): the artefacts under profiles/ -- kernel trace, four PMC passes, traffic JSON, update trace (tools/profile_step.sh), the driver's
# bench command in full, the long-window bench with its stamped timeline, the 8-rank plumbing run, and the whole GPU test suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
bash tools/profile_step.sh r05 --with-update > gpurun_out/final/profile_step.log 2>&1; echo "profile_step rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --timeline-out gpurun_out/final/step_timeline.txt > gpurun_out/final/bench_driver.json 2> gpurun_out/final/bench_driver.err; echo "bench driver rc=$?"
timeout 400 python bench.py --no-ppo --no-cpu-baseline --no-dropin --no-pmc-traffic --no-other-configs > gpurun_out/final/bench_long.json 2> gpurun_out/final/bench_long.err; echo "bench long rc=$?"
timeout 600 python bench.py --gpus 8 --same-gpu --dist-backend gloo --envs 512 --steps 20 --warmup 5 --no-cpu-baseline --no-worst-case --no-dropin --no-pmc-traffic --no-other-configs > gpurun_out/final/bench_8rank_gloo.json 2> gpurun_out/final/bench_8rank.err; echo "bench8 rc=$?"
timeout 300 python examples/train_ppo.py --updates 60 > gpurun_out/final/train_60_updates.jsonl 2> gpurun_out/final/train_60.err; echo "train60 rc=$?"; tail -1 gpurun_out/final/train_60_updates.jsonl | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/final/pytest.log
python - <<'PY'
import json
for f in ("bench_driver", "bench_long", "bench_8rank_gloo"):
    try:
        d = json.loads(open("gpurun_out/final/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "ppo", (d.get("ppo") or {}).get("samples_per_s"), (d.get("ppo") or {}).get("update_s"))
        print("   ", r.get("traffic_note"))
    except Exception as e:
        print(f, "ERR", e)
PY
