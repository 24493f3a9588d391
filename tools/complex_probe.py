"""Probe: the complex benchmark regex (floating window) and a window behind an unbounded prefix over random ASCII with
planted matches -- per-call time and (under rocprofv3 + tools/rocpd_timeline.py) the kernel timeline of the tails.
    python tools/complex_probe.py [bytes]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import rejit_amd
from rejit_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000_000
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
t = W.random_ascii_torch(n, 0xC0FFEE, dev)
rng = random.Random(7)
offs = W.plant_offsets(n, 80, 1000, seed=7)
for o in offs:
    W.plant(t, [o + 8], W.complex_regex_sample(rng))
for rx in (W.BENCH_REGEXES[3][0], "[a-z]+abcdefgh"):
    sc = rejit_amd.Scan(rejit_amd.Program(rx))
    for _ in range(3):
        k = sc.run(t.data_ptr(), n, stream=st)
    wall = []
    for _ in range(10):
        t0 = time.perf_counter()
        k = sc.run(t.data_ptr(), n, stream=st)
        wall.append(time.perf_counter() - t0)
    s = sc.stats()
    med = sorted(wall)[5]
    print(f"{rx}: {k} matches, call median {med * 1e3:.3f} ms (min {min(wall) * 1e3:.3f}), scan kernel {s['scan_ms']:.3f} ms, tails {med * 1e3 - s['scan_ms']:.3f} ms, "
          f"n / t_call = {n / med / 1e9:.0f} GB/s = {n / med / 8e12:.3f} of HBM peak; hits {s['n_hits']} candidates {s['n_candidates']}", flush=True)
