#!/usr/bin/env python3
"""Build profiles/rNN_pmc_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/profile_step.sh.

    python tools/mk_traffic.py <dir with pmc_1.txt pmc_2.txt pmc_1.log> <out.json>

The JSON is stamped with the sha256 of the kernel source it was measured on (csrc/hh_fused.hip); bench.py prints
`roofline.traffic` only when that stamp matches the source of the running build, otherwise null (a stale constant is worse
than no number)."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SRC = os.path.join(ROOT, "crowdnav_prediction_attngraph_amd", "csrc", "hh_fused.hip")


def kernel_stamp():
    return hashlib.sha256(open(KERNEL_SRC, "rb").read()).hexdigest()[:16]


def counter(path, name):
    for l in open(path):
        if "hh_fused_kernel" in l and (" " + name + " ") in l:
            p = l.split()
            return float(p[-1]), int(p[-3])
    raise SystemExit("no %s row for hh_fused_kernel in %s" % (name, path))


def main():
    d, out_path = sys.argv[1], sys.argv[2]
    fetch_kb, calls = counter(os.path.join(d, "pmc_1.txt"), "FETCH_SIZE")
    write_kb, _ = counter(os.path.join(d, "pmc_2.txt"), "WRITE_SIZE")
    line = [l for l in open(os.path.join(d, "pmc_1.log")) if l.startswith("{")][-1]
    b = json.loads(line)
    M = int(re.search(r"M=(\d+) live rows", b["roofline"]["kernel"]).group(1))
    alg = M * (2 + 256) * 4 + 3932160
    hbm = int(fetch_kb * 1024 * 2 + write_kb * 1024)
    out = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) -- python bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-ppo --no-worst-case",
        "kernel": "hh_fused_kernel", "kernel_source_sha16": kernel_stamp(), "launches_averaged": calls,
        "fetch_size_kb_per_launch": round(fetch_kb, 1), "write_size_kb_per_launch": round(write_kb, 1),
        "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section: 128-byte requests tallied at 64 B); WRITE_SIZE as reported",
        "hbm_bytes_per_launch_corrected": hbm, "algorithmic_bytes_per_launch": alg, "ratio": round(hbm / alg, 2),
        "live_rows_per_launch": M,
        "algorithmic_note": "live rows (mean of the timed steps of the profiled run) x (D=2 input floats + 256 output floats) + the 3.93 MB weight image once",
    }
    json.dump(out, open(out_path, "w"), indent=1)
    print(out["hbm_bytes_per_launch_corrected"], out["algorithmic_bytes_per_launch"], out["ratio"])


if __name__ == "__main__":
    main()
