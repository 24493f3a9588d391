"""Is verify_behind_in_regions' ~100 us fixed cost or work?  The same 5 GB scan with 0, 10, 100, 1000 planted hits
(run under rocprofv3 --kernel-trace --stats)."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import rejit_amd
from rejit_amd import workloads as W
n = 5_000_000_000
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
for planted in (0, 10, 100, 1000):
    t = W.random_ascii_torch(n, 0xC0FFEE + planted, dev)
    if planted:
        offs = W.plant_offsets(n, 80, planted, seed=7)
        W.plant(t, offs, b"qqqabcdefgh")
    sc = rejit_amd.Scan(rejit_amd.Program("[a-z]+abcdefgh"))
    for _ in range(2):
        sc.run(t.data_ptr(), n, stream=st)
    torch.cuda.synchronize()
    w = []
    for _ in range(5):
        t0 = time.perf_counter(); k = sc.run(t.data_ptr(), n, stream=st); w.append(time.perf_counter() - t0)
    print(planted, "planted:", k, "matches, call", round(sorted(w)[2] * 1e3, 3), "ms, scan", round(sc.stats()["scan_ms"], 3), flush=True)
    del t
