"""The bench line's contract (driver side: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline /
dtype / data / config, plus `roofline` and `cpu_baseline`) checked on the NEWEST committed record of the driver's command (profiles/rNN_bench.json): the
keys are there, the numbers are consistent with each other, and the self-explaining fractions follow from the same quantities.  Schema and internal
consistency only -- no performance threshold lives in a unit test (the record is a measurement artefact; bench.py itself is exercised on the GPU box by
tests/test_gpu_dist.py and by the driver)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    import glob
    import re
    cands = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")) if re.search(r"r\d+_bench\.json$", p))
    with open(cands[-1]) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_driver_line_has_the_contract_keys_and_consistent_numbers():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    E = d["config"]["envs_per_gpu"]
    assert abs(d["value"] - E * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) <= 2e-3 * d["value"]          # value = units of all ranks / time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["peak"] - 2500.0 / 3.0) < 0.1                                                             # dense bf16 / 3 split passes
    # the algorithmic FLOPs of the dominant kernel / its launch time
    rows = r["mean_detected_humans"] * E
    flops = 2.0 * rows * (128 * 512 + 512 * 1536 + 512 * 256)
    assert abs(flops / (r["launch_ms"] * 1e-3) / 1e12 - r["achieved"]) <= 0.01 * r["achieved"]
    # round 5: the fractions a reader could otherwise mistake `frac` for
    assert abs(r["frac_of_dense_bf16_algorithmic"] - r["achieved"] / 2500.0) < 1e-3
    assert 0.0 < r["whole_step_frac"] < r["frac"]
    assert r["traffic"] is None or r["traffic"] > rows * 258 * 4                                          # at least the algorithmic bytes
    assert set(("env_step", "orca_lane", "hh_fused", "rn_fused", "step")) <= set(r["step_decomposition_us"])
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    legs = d["other_baseline_configs_1gpu"]
    assert len(legs) == 3 and all(("env_steps_per_s" in x) != ("error" in x) for x in legs)            # a leg is a number or says why not
    assert "configs[2]" in legs[0]["config"] and "configs[3]" in legs[1]["config"] and "configs[4]" in legs[2]["config"]


def test_bench_cli_parses_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--no-other-configs" in out.stdout
