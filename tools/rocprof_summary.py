#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as text: per-kernel calls / total / avg / min / max
and the launch geometry + register/LDS footprint.  usage: rocprof_summary.py <results.db> [more.db ...]"""
import sqlite3, sys

for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    print(f"# {path}")
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'grid':>9s} {'wg':>5s} {'lds':>7s} {'vgpr':>5s} {'sgpr':>5s}")
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                     "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(sgpr_count) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    for r in rows:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]:11.1f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f} {r[6]:9d} {r[7]:5d} {r[8]:7d} {r[9]:5d} {r[10]:5d}")
    print(f"# total kernel time {tot:.1f} us")
