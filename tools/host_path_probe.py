#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-text entry points (rj_match_all on a host buffer)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rejit_amd
from rejit_amd import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
t = W.random_ascii_numpy(n, seed=3)
W.plant(t, W.plant_offsets(n, 6, 1000, seed=3), b"regexp")
tb = t.tobytes()
p = rejit_amd.Program(b"regexp")
p.count(tb)
for _ in range(3):
    t0 = time.perf_counter(); c = p.count(tb); dt = time.perf_counter() - t0
    print(f"rj_match_all(count) over {n/1e9:.1f} GB of host text: {dt*1e3:.1f} ms = {n/dt/1e9:.1f} GB/s, {c} matches")
t0 = time.perf_counter(); f = p.match_first(tb); dt = time.perf_counter() - t0
print(f"rj_match_first: {dt*1e3:.3f} ms -> {f}")
