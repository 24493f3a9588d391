#!/usr/bin/env python3
"""Per-pattern device timings (HIP events inside librejit_hip) for the bench workloads.
Usage on the GPU box:  python tools/perf_probe.py [fasta_n] [literal_bytes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rejit_amd
from rejit_amd import workloads as W

fasta_n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
lit_n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
dev = torch.device("cuda:0")
text = W.fasta_stripped_torch(fasta_n, dev)
n = text.numel()
st = torch.cuda.current_stream(dev).cuda_stream
for rx in W.REGEXDNA_PATTERNS:
    p = rejit_amd.Program(rx); sc = rejit_amd.Scan(p)
    for _ in range(3): sc.run(text.data_ptr(), n, stream=st)
    best = None
    for _ in range(5):
        t0 = time.perf_counter(); c = sc.run(text.data_ptr(), n, stream=st); dt = time.perf_counter() - t0
        s = sc.stats(); s["wall_ms"] = dt * 1e3
        if best is None or s["scan_ms"] < best["scan_ms"]: best = s
    info = p.info()
    print(f"{rx:32s} K={info['n_windows']} hits={best['n_hits']:8d} cands={best['n_candidates']:7d} scan={best['scan_ms']:.3f}ms "
          f"({n/best['scan_ms']/1e6:7.1f} GB/s) dev_total={best['total_ms']:.3f}ms wall={best['wall_ms']:.3f}ms large={best['large_path']}")
del text
t = W.random_ascii_torch(lit_n, 1, dev)
for rx in ["regexp", "rege", "abcdefgh", "(alternation|strings)", "[0-9]+x"]:
    p = rejit_amd.Program(rx); sc = rejit_amd.Scan(p)
    for _ in range(3): sc.run(t.data_ptr(), lit_n, stream=st)
    t0 = time.perf_counter(); c = sc.run(t.data_ptr(), lit_n, stream=st); dt = time.perf_counter() - t0
    s = sc.stats(); info = p.info()
    print(f"{rx:32s} mode={info['scan_mode']} K={info['n_windows']} len={info['window_len']} hits={s['n_hits']:9d} matches={c:8d} "
          f"scan={s['scan_ms']:.3f}ms ({lit_n/s['scan_ms']/1e6:7.1f} GB/s) dev_total={s['total_ms']:.3f}ms wall={dt*1e3:.3f}ms")
