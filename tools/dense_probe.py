#!/usr/bin/env python3
"""Patterns without a strong fast-forward window over random ASCII (and over text with a line break every
61 bytes for the `^` family): device timings per run.  usage: dense_probe.py [bytes] [regex ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rejit_amd
from rejit_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
rxs = sys.argv[2:] or ["[0-9]+x", "[a-f]+[0-9]", "x[0-9]*y", "(ab|cd)+e", "[A-Z][a-z]+ [A-Z][a-z]+", "[a-z]+", "[0-9]+", "[0-9][0-9][0-9]",
                       "^", "$", "^[a-z]+:", "^[A-Z]", "[@#]", "\\n"]
dev = torch.device("cuda:0")
t = W.random_ascii_torch(n, 1, dev)
tl = t.clone()
tl[torch.arange(60, n, 61, device=dev)] = 10      # a line break every 61 bytes
st = torch.cuda.current_stream(dev).cuda_stream
for rx in rxs:
    text = tl if (rx.startswith("^") or rx in ("$", "\\n")) else t
    p = rejit_amd.Program(rx); sc = rejit_amd.Scan(p)
    for _ in range(2): sc.run(text.data_ptr(), n, stream=st)
    t0 = time.perf_counter(); c = sc.run(text.data_ptr(), n, stream=st); dt = time.perf_counter() - t0
    s = sc.stats(); info = p.info()
    print(f"{rx:28s} mode={info['scan_mode']} W={info['n_words']} hits={s['n_hits']:10d} cands={s['n_candidates']:9d} matches={c:9d} "
          f"scan={s['scan_ms']:.3f}ms total={s['total_ms']:.3f}ms ({n/s['total_ms']/1e6:7.1f} GB/s) large={s['large_path']} linear={s['linear_path']} retries={s['retries']}", flush=True)
