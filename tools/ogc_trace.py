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
NAMES = {0: "first workgroup starts", 1: "counts of all earlier regions + own entries read (last workgroup)", 2: "nearest earlier end read",
         3: "copied and checked", 4: "last block: host counters"}
for rep in range(5):
    lib.rj_debug_trace_reset()
    k = sc.run(t.data_ptr(), n, stream=st)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    lib.rj_debug_trace(buf)
    if rep >= 3:
        print(k, "matches; call", round(sc.stats()["total_ms"], 3), "scan", round(sc.stats()["scan_ms"], 3))
        for i in sorted(NAMES):
            print("   %-70s %+8.2f us" % (NAMES[i], (buf[i] - buf[0]) / 100.0))
