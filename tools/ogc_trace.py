#!/usr/bin/env python3
"""Phase stamps of offsets_gather_check (debug library of tools/verify_trace.py --build): literal `regexp` over 5 GB,
1000 hits.  Stamp 0 = the earliest workgroup's start, the others the latest workgroup to pass."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from rejit_amd import api
import torch
api.LIB = os.path.join(api.PKG, "librejit_hip_trace.so")
import rejit_amd
from rejit_amd import workloads as W
lib = api.load_library()
lib.rj_debug_trace.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
n = 5_000_000_000
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
t = W.random_ascii_torch(n, 0xC0FFEE, dev)
W.plant(t, W.plant_offsets(n, 6, 1000, seed=7), b"regexp")
sc = rejit_amd.Scan(rejit_amd.Program("regexp"))
lib.rj_debug_trace_wide.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
NAMES = ["workgroup starts", "counts of all earlier regions summed (thread 0)", "own count + entries read, wave sums exchanged", "nearest earlier end read",
         "copied and checked", "last block: host counters written"]
for rep in range(5):
    k = sc.run(t.data_ptr(), n, stream=st)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * 4096))()
    lib.rj_debug_trace_wide(buf)
    if rep >= 3:
        blocks = [b for b in range(4096) if buf[8 * b]]
        t0 = min(buf[8 * b] for b in blocks)
        print(k, "matches; call", round(sc.stats()["total_ms"], 3), "scan", round(sc.stats()["scan_ms"], 3), "; workgroups", len(blocks))
        for i, name in enumerate(NAMES):
            vals = [(buf[8 * b + i] - t0) / 100.0 for b in blocks if buf[8 * b + i]]
            if vals:
                print("   %-62s first %+7.2f  last %+7.2f us (workgroup %d)" % (name, min(vals), max(vals), max(blocks, key=lambda b: buf[8 * b + i])))
