#!/usr/bin/env python3
"""GPU: where a headline step's host time goes -- rj_multi_start and rj_multi_finish timed separately in the two-in-flight loop
(tails on own streams), and the same loop with THREE objects in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rejit_amd
from rejit_amd import workloads as W
dev = torch.device("cuda:0")
progs = [rejit_amd.Program(p) for p in W.REGEXDNA_PATTERNS]
main = torch.cuda.current_stream(dev).cuda_stream
text = W.fasta_stripped_torch(50_000_000, dev)
n = int(text.numel())
for depth in (2, 3, 4):
    ms = [rejit_amd.MultiScan(progs) for _ in range(depth)]
    for m in ms:
        m.set_tail_stream(True)
        m.run(text.data_ptr(), n, stream=main)
    torch.cuda.synchronize(dev)
    steps = 60
    t_start = t_fin = 0.0
    busy = [False] * depth
    t0 = time.perf_counter()
    for k in range(steps):
        j = k % depth
        if busy[j]:
            a = time.perf_counter(); ms[j].finish(); t_fin += time.perf_counter() - a
        a = time.perf_counter(); ms[j].start(text.data_ptr(), n, stream=main); t_start += time.perf_counter() - a
        busy[j] = True
    for j in range(depth):
        jj = (steps + j) % depth
        if busy[jj]:
            ms[jj].finish()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    print(f"{depth} in flight: {dt * 1e3:.4f} ms/step; host: start {t_start / steps * 1e6:.1f} us, finish (incl. waiting) {t_fin / steps * 1e6:.1f} us per step", flush=True)
