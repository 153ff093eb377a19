"""Digest of profiles/summarize.py PMC outputs: for the kernels that take >= --min-pct of the traced time, one block per kernel with
its per-launch counters and the ratios DESIGN.md quotes (clock, MFMA-busy, L2 request latency, bytes).  Counter conventions on gfx950
as profiles/README.md has them: SQ_* are summed over the dispatch records of 32 shader engines, SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count in units of 4 cycles, SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE is per XCD.
  python tools/pmc_digest.py pmc_2.txt pmc_3.txt pmc_4.txt [--min-pct 1.5]"""
import re
import sys
from collections import defaultdict

args = sys.argv[1:]
min_pct = 1.5
if "--min-pct" in args:
    i = args.index("--min-pct")
    min_pct = float(args[i + 1])
    del args[i:i + 2]
files = args
dur, calls, ctr, heads = {}, {}, defaultdict(dict), []
for f in files:
    pmc = False
    for ln in open(f):
        if ln.startswith("# pmc pass"):
            heads.append(ln.strip())
        if ln.startswith("# PMC"):
            pmc = True
            continue
        m = re.match(r"(\S+)\s+(\d+,\d+)\s+(.*)", ln)
        if not m or ln.startswith("kernel"):
            continue
        key = (m.group(1), m.group(2))
        rest = m.group(3).split()
        if not pmc and len(rest) >= 7:
            if float(rest[6]) >= min_pct:
                dur.setdefault(key, float(rest[3])); calls.setdefault(key, int(rest[1]))
        elif pmc and len(rest) == 4:
            ctr[key][rest[0]] = (int(rest[1]), float(rest[2]))
for h in heads:
    print(h)
print()
for key in sorted(dur, key=lambda k: -dur[k] * calls[k]):
    c, n, us = ctr[key], calls[key], dur[key]
    per = {k: v[1] / n for k, v in c.items()}  # per launch, all records summed
    print("%s  grid %s  calls %d  avg %.1f us" % (key[0][:78], key[1], n, us))
    out = []
    if "GRBM_GUI_ACTIVE" in per:
        cyc = per["GRBM_GUI_ACTIVE"] / 8.0
        out.append("clock %.2f GHz (GRBM_GUI_ACTIVE %.3g cycles per launch and XCD)" % (cyc / us / 1e3, cyc))
    if "SQ_BUSY_CYCLES" in per and "SQ_VALU_MFMA_BUSY_CYCLES" in per:
        busy = per["SQ_BUSY_CYCLES"] / 32.0
        out.append("MFMA-busy %.0f %% of the SIMD cycles (%.3g of %.3g)" % (100 * per["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / busy, per["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0, busy))
        if "SQ_WAVE_CYCLES" in per:
            out.append("resident wavefronts %.0f (SQ_WAVE_CYCLES x 4 / busy cycles)" % (per["SQ_WAVE_CYCLES"] * 4 / busy))
            for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
                if k in per:
                    out.append("%s / SQ_WAVE_CYCLES %.2f" % (k, per[k] / per["SQ_WAVE_CYCLES"]))
    if "SQ_INSTS_MFMA" in per:
        out.append("MFMA instructions %.4g per launch" % per["SQ_INSTS_MFMA"])
    if "TCP_TCC_READ_REQ_sum" in per and per["TCP_TCC_READ_REQ_sum"] > 0:
        out.append("L1 -> L2 read requests %.4g per launch, mean latency %.0f cycles" % (per["TCP_TCC_READ_REQ_sum"], per.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / per["TCP_TCC_READ_REQ_sum"]))
    if "TCC_REQ_sum" in per:
        out.append("L2 requests %.4g per launch, fabric reads (TCC_EA0_RDREQ) %.4g" % (per["TCC_REQ_sum"], per.get("TCC_EA0_RDREQ_sum", 0)))
    if "SQ_LDS_BANK_CONFLICT" in per and per.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        out.append("LDS bank conflicts %.1f %% of the LDS cycles" % (100 * per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]))
    for o in out:
        print("    " + o)
    print()
