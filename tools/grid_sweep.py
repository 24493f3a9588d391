#!/usr/bin/env python3
"""Scan-kernel time of one regexdna pattern for several grid sizes (RJ_SCAN_GRID is read once per
process, so each size runs in its own subprocess).  usage: grid_sweep.py [fasta_n] [grids...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import torch, rejit_amd
from rejit_amd import workloads as W
dev = torch.device("cuda:0")
text = W.fasta_stripped_torch(int(sys.argv[1]), dev); n = text.numel()
st = torch.cuda.current_stream(dev).cuda_stream
for rx in ["agggtaaa|tttaccct", "agg[act]taaa|ttta[agt]cct"]:
    sc = rejit_amd.Scan(rejit_amd.Program(rx))
    for _ in range(3): sc.run(text.data_ptr(), n, stream=st)
    v = sorted(( (sc.run(text.data_ptr(), n, stream=st), sc.stats())[1] for _ in range(9)), key=lambda s: s["scan_ms"])
    m = v[len(v)//2]
    print("grid=%%-6s %%-28s scan=%%.4f ms (%%.0f GB/s) total=%%.4f" %% (os.environ.get("RJ_SCAN_GRID","auto"), rx, m["scan_ms"], n/m["scan_ms"]/1e6, m["total_ms"]))
''' % ROOT
n = sys.argv[1] if len(sys.argv) > 1 else "50000000"
grids = sys.argv[2:] or ["0", "1024", "2048", "3072", "4096", "6144", "8192", "16384"]
for g in grids:
    env = dict(os.environ)
    if g != "0": env["RJ_SCAN_GRID"] = g
    else: env.pop("RJ_SCAN_GRID", None)
    subprocess.run([sys.executable, "-c", CHILD, n], env=env, check=False)
