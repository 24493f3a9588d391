#!/usr/bin/env python3
"""Per-dispatch timeline of a rocprofv3 --kernel-trace rocpd database: start offset, duration, kernel.
usage: rocpd_timeline.py results.db [max_rows]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
start = "start" if "start" in cols else [c for c in cols if "start" in c.lower()][0]
end = "end" if "end" in cols else [c for c in cols if "end" in c.lower()][0]
rows = db.execute(f'select name, "{start}", "{end}" from {view} order by "{start}"').fetchall()
t0 = rows[0][1] if rows else 0
prev_end = t0
for name, s, e in rows[:limit]:
    short = re.sub(r"\(.*", "", name).replace("rejit_amd::", "").replace("void ", "")
    print(f"{(s - t0) / 1e3:12.1f} us  gap {(s - prev_end) / 1e3:9.1f}  dur {(e - s) / 1e3:10.1f}  {short[:90]}")
    prev_end = e
