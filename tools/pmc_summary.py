#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc <COUNTER> --output-format csv run: per kernel calls and the
avg/min/max counter value.  usage: pmc_summary.py <dir> <COUNTER>"""
import csv, glob, os, re, sys
from collections import defaultdict

d, counter = sys.argv[1], sys.argv[2]
rows = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
def clusters(v):
    """launches of one kernel at several text sizes: the values grouped within 3 %, as mean x count (largest first)"""
    out = []
    for x in sorted(v, reverse=True):
        if out and x >= out[-1][0] / out[-1][1] * 0.97:
            out[-1][0] += x
            out[-1][1] += 1
        else:
            out.append([x, 1])
    return " ".join(f"{s / c:.1f}x{c}" for s, c in out[:6])


print(f"{'kernel':60s} {'calls':>6s} {'avg':>14s} {'min':>14s} {'max':>14s}  groups (mean x launches)")
for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    short = re.sub(r"\(.*", "", name).replace("rejit_amd::", "").replace("void ", "")
    if "at::native" in short or "rocclr" in short:
        continue
    print(f"{short[:60]:60s} {len(v):6d} {sum(v)/len(v):14.1f} {min(v):14.1f} {max(v):14.1f}  {clusters(v)}")
