#!/usr/bin/env python3
"""Many-small-files shape (BASELINE C5 on one GPU): rj_match_all_batch vs one rj_match_all per file
vs the real reference on one host core.  usage: batch_probe.py [n_files] [avg_bytes]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import rejit_amd
from rejit_amd import workloads as W

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
avg = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
rng = np.random.default_rng(1)
sizes = rng.integers(avg // 4, avg * 7 // 4, n_files)
big = W.random_ascii_numpy(int(sizes.sum()), seed=5)
offs = np.concatenate([[0], np.cumsum(sizes)])
W.plant(big, W.plant_offsets(big.size, 6, n_files // 4, seed=9), b"regexp")
files = [big[offs[i]:offs[i + 1]].tobytes() for i in range(n_files)]
total = sum(len(f) for f in files)
for rx in (b"regexp", b"(regexp|abcdefgh) [a-z]"):
    p = rejit_amd.Program(rx)
    p.match_all_batch(files)   # warm: pinned staging + device buffers are grow-only
    t0 = time.perf_counter(); res = p.match_all_batch(files); t_batch = time.perf_counter() - t0
    sub = files[:2000]
    t0 = time.perf_counter(); one = [p.match_all(f) for f in sub]; t_loop = (time.perf_counter() - t0) * n_files / len(sub)
    assert res[:len(sub)] == one
    line = f"{rx.decode():28s} {n_files} files, {total/1e9:.2f} GB: batch {t_batch*1e3:8.1f} ms ({total/t_batch/1e9:6.2f} GB/s)   per-file loop {t_loop*1e3:9.1f} ms ({total/t_loop/1e9:6.2f} GB/s, extrapolated from {len(sub)})"
    try:
        from checkers import Ref
        ref = Ref(use_ff=1, ff_reduce=0)
        t0 = time.perf_counter(); rr = [ref.match_all(rx, f) for f in sub[:500]]; t_ref = (time.perf_counter() - t0) * n_files / 500
        assert rr == one[:500]
        line += f"   reference 1 core {t_ref*1e3:9.1f} ms ({total/t_ref/1e9:6.2f} GB/s)"
    except Exception as e:  # the reference library is not present
        line += f"   (reference not available: {e})"
    print(line)

# the C call alone (argument arrays prebuilt), to separate the Python marshalling from the library
import ctypes
lib = rejit_amd.load_library()
k = len(files)
arr = (ctypes.c_char_p * k)(*files)
sz = (ctypes.c_size_t * k)(*[len(f) for f in files])
counts = (ctypes.c_uint64 * k)()
p = rejit_amd.Program(b"regexp")
for _ in range(3):
    spans = ctypes.POINTER(ctypes.c_uint64)()
    t0 = time.perf_counter()
    tot = lib.rj_match_all_batch(p._h, arr, sz, k, counts, ctypes.byref(spans))
    dt = time.perf_counter() - t0
    lib.rj_free_spans(spans)
    print(f"rj_match_all_batch C call alone: {dt*1e3:.1f} ms = {total/dt/1e9:.1f} GB/s ({tot} matches)")
