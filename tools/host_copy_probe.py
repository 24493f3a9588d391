#!/usr/bin/env python3
"""Large HOST buffers through the library's own staging (host_api.hip: staged_upload / staged_download, round 6) against the
runtime's plain copy (RJ_NO_STAGED_COPY=1 in a child process) and against a bare pageable hipMemcpy of the same bytes:
MatchAllCount of a literal over a pageable host text, and ReplaceAll (text up, new text down).
usage: host_copy_probe.py [bytes]"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000_000
child = len(sys.argv) > 2
import numpy as np
import torch
import rejit_amd
from rejit_amd import workloads as W
t = W.random_ascii_torch(n, 3, torch.device("cuda:0")).cpu().numpy()
W.plant(t, W.plant_offsets(n, 6, 1000, seed=3), b"regexp")
tb = t.tobytes()
del t
tag = "plain hipMemcpyAsync" if os.environ.get("RJ_NO_STAGED_COPY") else "staged (pinned ring + copy pool)"
p = rejit_amd.Program(b"regexp")
p.count(tb)
best = min(_timed for _timed in [(lambda: (time.perf_counter(), p.count(tb), time.perf_counter()))() for _ in range(4)] for _timed in [_timed[2] - _timed[0]])
print(f"{tag}: rj_match_all(count) over {n/1e9:.1f} GB of pageable host text: {best*1e3:.1f} ms = {n/best/1e9:.1f} GB/s")
q = rejit_amd.Program(b"regexp")
small = tb[: min(n, 1_000_000_000)]
q.replace_all(small, b"REGEXP!")
t0 = time.perf_counter(); k, out = q.replace_all(small, b"REGEXP!"); dt = time.perf_counter() - t0
print(f"{tag}: rj_replace_all over {len(small)/1e9:.1f} GB (text up, new text down): {dt*1e3:.1f} ms = {2*len(small)/dt/1e9:.1f} GB/s both ways, {k} matches")
if not child:
    # the bare copy of the same pageable bytes
    d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    for _ in range(3):
        t0 = time.perf_counter(); rc = hip.hipMemcpy(d.data_ptr(), tb, n, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"bare hipMemcpy H2D of the same pageable bytes: {dt*1e3:.1f} ms = {n/dt/1e9:.1f} GB/s (rc {rc})")
    env = dict(os.environ, RJ_NO_STAGED_COPY="1")
    subprocess.call([sys.executable, __file__, str(n), "child"], env=env)
