#!/usr/bin/env python3
"""Where does a verify kernel's per-hit latency go?  A DEBUG build of the library (kernels.hip compiled with
-DRJ_TRACE_VERIFY: wall-clock stamps of the phases, 10 ns units) run on a 1 GB text with ONE planted hit.
   python tools/verify_trace.py --build     here (no GPU): rejit_amd/librejit_hip_trace.so
   python tools/verify_trace.py             on the GPU box
The product library never contains the stamps."""
import ctypes, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from rejit_amd import api

TRACE_LIB = os.path.join(api.PKG, "librejit_hip_trace.so")
if "--build" in sys.argv:
    api.build()
    objdir = os.path.join(api.PKG, "build")
    traced = ("kernels.hip", "verify_lds.hip", "plane_scan.hip")   # (round 6: the window scans, scan_dense_walk and the selection kernels left kernels.hip; none of them carries stamps)
    objs = []
    for src in traced:
        objs.append(os.path.join(objdir, os.path.splitext(src)[0] + "_trace.o"))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-DRJ_TRACE_VERIFY", "-c",
                               os.path.join(api.CSRC, src), "-o", objs[-1]])
    objs += [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in api.SOURCES if s not in traced]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", TRACE_LIB] + objs)
    print(TRACE_LIB)
    sys.exit(0)

import torch
api.LIB = TRACE_LIB
import rejit_amd
from rejit_amd import workloads as W
lib = api.load_library()
lib.rj_debug_trace.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
lib.rj_debug_trace_lds.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
# RJ_NO_LDS_WALK=1 in the environment: the old kernels (kernels.hip); else the LDS walkers (verify_lds.hip)
read_trace = lib.rj_debug_trace if os.environ.get("RJ_NO_LDS_WALK") else lib.rj_debug_trace_lds
reset_trace = lib.rj_debug_trace_reset if os.environ.get("RJ_NO_LDS_WALK") else lib.rj_debug_trace_lds_reset
n = 1 << 30
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
NAMES = {0: "kernel start (wg 0)", 1: "tables staged", 2: "count read", 3: "hit read", 4: "window tested (text read)", 5: "forward from the cut done",
         6: "left-most start found", 7: "longest end found", 8: "ballot", 9: "stored", 10: "wg 0 done"}
rng = __import__("random").Random(5)
# (stamp 0 = the earliest lane, the others the latest: one hit gives its chain, a thousand the slowest hit's)
for rx, plant, hits in (("[a-z]+abcdefgh", b"0qqqabcdefgh0", 1), ("[a-z]+abcdefgh", b"0" + b"q" * 40 + b"abcdefgh0", 1),
                        (W.BENCH_REGEXES[3][0], None, 1), ("[a-z]+abcdefgh", None, 200), (W.BENCH_REGEXES[3][0], None, 200)):
    t = W.random_ascii_torch(n, 0xC0FFEE, dev)
    for o in ([n // 2 + 5] if hits == 1 else W.plant_offsets(n, 80, hits, seed=7)):
        W.plant(t, [o + 8], plant if plant is not None else W.complex_regex_sample(rng))
    sc = rejit_amd.Scan(rejit_amd.Program(rx))
    for rep in range(4):
        reset_trace()
        k = sc.run(t.data_ptr(), n, stream=st)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 64)()
        assert read_trace(buf) == 0
        if rep >= 2:
            t0 = buf[0]
            print(rx[:30], hits, "planted;", k, "matches; call", round(sc.stats()["total_ms"], 3), "ms, scan", round(sc.stats()["scan_ms"], 3))
            for i in sorted(NAMES):
                if buf[i]:
                    print("   %-28s %+8.2f us" % (NAMES[i], (buf[i] - t0) / 100.0))
    del t
