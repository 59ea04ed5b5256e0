#!/usr/bin/env python3
"""Per-kernel summary (calls, avg/min/max/total ns, registers, LDS, scratch) from a rocprofv3 rocpd
SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db)."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start), "
         "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size), "
         "max(d.grid_size_x), max(d.grid_size_y), max(d.workgroup_size_x) "
         "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 6 desc" % (disp, sym))
    rows = list(c.execute(q))
    total = sum(r[5] for r in rows) or 1
    print("%-8s %12s %12s %12s %14s %6s  %5s %5s %5s %7s %7s  %s" %
          ("calls", "avg_ns", "min_ns", "max_ns", "total_ns", "pct", "vgpr", "agpr", "sgpr", "lds_B", "scr_B", "kernel (grid, block)"))
    for r in rows:
        print("%-8d %12.0f %12d %12d %14d %6.2f  %5d %5d %5d %7d %7d  %s (%dx%d, %d)" %
              (r[1], r[2], r[3], r[4], r[5], 100.0 * r[5] / total, r[6], r[7], r[8], r[9], r[10], r[0], r[11], r[12], r[13]))


if __name__ == "__main__":
    main(sys.argv[1])
