#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the text summaries kept under profiles/.

    python profiles/summarize.py gpurun_out/prof_r01/r01_results.db "header text" > profiles/r01_xxx.txt
Groups dispatches by (kernel, grid) so the QKV projection GEMM is separated from the small GEMMs that share its
template instantiation.  With --pmc data present, per-kernel counter sums / per-dispatch averages are appended.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    header = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = list(cur.execute(
        "select s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, count(*), sum(d.end - d.start), avg(d.end - d.start), "
        "min(d.end - d.start), max(d.end - d.start), s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count, d.group_segment_size "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name, d.grid_size_x, d.grid_size_y order by 6 desc"))
    total = sum(r[5] for r in rows) or 1
    print("# " + header)
    print("%-74s %13s %5s %7s %12s %10s %10s %10s %6s %5s %5s %5s %7s" % ("kernel", "grid(x,y)", "wg", "calls", "total_us", "avg_us", "min_us", "max_us",
                                                                         "pct", "vgpr", "agpr", "sgpr", "lds"))
    for r in rows:
        name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0] if not name.startswith("gemm_nt") else name.split("(")[0]
        print("%-74s %13s %5d %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %5d %7d" % (
            name[:74], "%d,%d" % (r[1], r[2]), r[3], r[4], r[5] / 1e3, r[6] / 1e3, r[7] / 1e3, r[8] / 1e3, 100.0 * r[5] / total, r[9], r[10], r[11], r[12]))
    try:
        pmc = list(cur.execute(
            "select s.kernel_name, d.grid_size_x, d.grid_size_y, i.name, count(*), sum(p.value), avg(p.value) "
            "from rocpd_pmc_event p join rocpd_info_pmc i on p.pmc_id = i.id "
            "join rocpd_kernel_dispatch d on p.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
            "group by s.kernel_name, d.grid_size_x, d.grid_size_y, i.name order by 6 desc"))
    except sqlite3.Error as e:  # schema without counters
        pmc = []
    if pmc:
        print("\n# PMC counters (sum and per-dispatch average)")
        print("%-74s %13s %-28s %7s %16s %16s" % ("kernel", "grid(x,y)", "counter", "calls", "sum", "avg/dispatch"))
        for r in pmc:
            name = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            print("%-74s %13s %-28s %7d %16.1f %16.1f" % (name[:74], "%d,%d" % (r[1], r[2]), r[3], r[4], r[5], r[6]))


if __name__ == "__main__":
    main()
