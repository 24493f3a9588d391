#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats rocpd database (sqlite) as text:
per-kernel calls / total / average / min / max duration.   usage: prof_summary.py results.db"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s}")
for name, calls, tot, avg, mn, mx, vg, sg, lds in rows:
    short = re.sub(r"\(.*", "", name)
    short = short.replace("rejit_amd::", "")
    print(f"{short[:70]:70s} {calls:6d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f} {vg:5d} {sg:5d} {lds:6d}")
