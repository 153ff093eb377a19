#!/usr/bin/env python3
"""Print the kernel timeline of one rollout step (env_step -> ... -> next env_step) from a rocprofv3 kernel-trace database:
start / end / duration / queue of every dispatch, so that launch gaps and side-stream overlap can be read off.

    python tools/step_timeline.py <results.db>
"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
# find the k-th hh_fused dispatch and print everything from the env_step before it to the env_step after it
idx = [i for i, r in enumerate(rows) if "hh_fused" in r[0]]
k = idx[len(idx) * 3 // 4]
lo = k
while lo > 0 and "env_step" not in rows[lo][0]: lo -= 1
hi = k
while hi < len(rows) - 1 and "env_step" not in rows[hi][0]: hi += 1
hi = min(hi + 4, len(rows) - 1)
t0 = rows[lo][1]
for r in rows[lo:hi + 1]:
    name = r[0].replace("_ZN12_GLOBAL__N_1", "").split("E")[0][:40]
    print("%8.1f -> %8.1f us  (%6.1f)  q%-3d grid %-8d %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4], r[0][:60]))
