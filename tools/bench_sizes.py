#!/usr/bin/env python3
"""Throughput vs text size, the reference harness's protocol (tools/benchmarks/engines/
bench_engine.cc:177-249: random text in [low,high), sizes 2^3..2^21, bytes/s = size / time,
compile excluded) extended to GPU-sized inputs.  Device-resident text; `cpu` columns = the real
reference (oracle/_ref, default flags) on one host core where the prebuilt library exists.

    python tools/bench_sizes.py [all | regex_index ...]      # indices into workloads.BENCH_REGEXES (all: the twelve of run.py:347-360)
    BENCH_SIZES_TIMING=1: with the scan kernel's start event (6-9 us per call), as rounds 2-4 measured
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rejit_amd
from rejit_amd import workloads as W
import checkers

dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
ref = checkers.Ref(use_ff=1, ff_early=1, ff_reduce=0, parser_opt=1) if checkers.have_ref() else None
sizes = [1 << k for k in range(3, 22, 3)] + [1 << 16, 1 << 20, 1 << 22, 1 << 24, 1 << 27, 1 << 30]
sizes = sorted(set(sizes))
# the C ABI's default: no start event on the scan kernel (rj_scan_set_timing; the Python binding switches it on for scan_ms)
TIMED = os.environ.get("BENCH_SIZES_TIMING", "0") == "1"
which = list(range(len(W.BENCH_REGEXES))) if sys.argv[1:] == ["all"] else ([int(a) for a in sys.argv[1:]] or [1, 3, 4, 11])
print("%-58s %10s %12s %12s %12s" % ("regexp", "size", "gpu_us/call", "gpu_GB/s", "cpu1_GB/s"))
for idx in which:
    rx, lo, hi = W.BENCH_REGEXES[idx]
    prog = rejit_amd.Program(rx); sc = rejit_amd.Scan(prog); sc.set_timing(TIMED)
    big = W.random_ascii_torch(max(sizes), 42 + idx, dev, ord(lo), ord(hi))
    for n in sizes:
        for _ in range(3): sc.run(big.data_ptr(), n, stream=st)
        reps = 20 if n < (1 << 24) else 5
        t0 = time.perf_counter()
        for _ in range(reps): sc.run(big.data_ptr(), n, stream=st)
        dt = (time.perf_counter() - t0) / reps
        cpu = ""
        if ref is not None and n <= (1 << 27):
            host = big[:n].cpu().numpy().tobytes()
            it = max(1, min(50, (1 << 26) // n))
            t0 = time.perf_counter(); ref.lib.ref_match_all_repeat(rx.encode(), host, n, it); c = (time.perf_counter() - t0) / it
            cpu = "%.3f" % (n / c / 1e9)
        print("%-58s %10d %12.1f %12.3f %12s" % (rx[:58], n, dt * 1e6, n / dt / 1e9, cpu))
    del big
