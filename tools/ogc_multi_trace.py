#!/usr/bin/env python3
"""Per-workgroup phase stamps of offsets_gather_check_multi (debug library of tools/verify_trace.py --build): the nine
regexdna patterns over the 500 MB text.  (A slot is shared by the nine patterns' workgroups of one index: the last writer wins.)"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from rejit_amd import api
import torch
api.LIB = os.path.join(api.PKG, "librejit_hip_trace.so")
import rejit_amd
from rejit_amd import workloads as W
lib = api.load_library()
lib.rj_debug_trace_wide.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
t = W.fasta_stripped_torch(50_000_000, dev)
m = rejit_amd.MultiScan([rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS])
NAMES = ["workgroup starts", "own counts + entries read, wave sums exchanged", "granules of the workgroups before read", "nearest earlier end read", "copied and checked",
         "last block: host counters written"]
for rep in range(5):
    c = m.run(t.data_ptr(), t.numel(), stream=st)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * 4096))()
    lib.rj_debug_trace_wide(buf)
    if rep >= 3:
        blocks = [b for b in range(4096) if buf[8 * b]]
        t0 = min(buf[8 * b] for b in blocks)
        print(sum(c), "matches; workgroup indices", len(blocks))
        for i, name in enumerate(NAMES):
            vals = sorted((buf[8 * b + i] - t0) / 100.0 for b in blocks if buf[8 * b + i] >= t0)
            if vals:
                print("   %-52s first %+7.2f  median %+7.2f  last %+7.2f us" % (name, vals[0], vals[len(vals) // 2], vals[-1]))
