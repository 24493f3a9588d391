#!/usr/bin/env python3
"""Per-call latency of small texts (the unchanged jrep does one MatchAll per file): device-resident and host
text, a few sizes.  usage: small_probe.py [regex] [size ...]   (run under rocprofv3 --kernel-trace --stats with
ONE size to see the kernel's share of the call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rejit_amd
from rejit_amd import workloads as W

rx = sys.argv[1] if len(sys.argv) > 1 else "regexp"
sizes = [int(a) for a in sys.argv[2:]] or [8, 512, 4096, 20000, 32768, 65536]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
big = W.random_ascii_torch(1 << 20, 7, dev)
host_big = big.cpu().numpy().tobytes()
p = rejit_amd.Program(rx); sc = rejit_amd.Scan(p)
print("%-20s %8s %14s %14s" % ("regexp", "size", "device_us/call", "host_us/call"))
for n in sizes:
    for _ in range(50): sc.run(big.data_ptr(), n, stream=st)
    t0 = time.perf_counter()
    for _ in range(500): sc.run(big.data_ptr(), n, stream=st)
    d_us = (time.perf_counter() - t0) / 500 * 1e6
    host = host_big[:n]
    lib = rejit_amd.load_library()
    for _ in range(50): lib.rj_match_all(p._h, host, n, None)
    t0 = time.perf_counter()
    for _ in range(500): lib.rj_match_all(p._h, host, n, None)
    h_us = (time.perf_counter() - t0) / 500 * 1e6
    print("%-20s %8d %14.1f %14.1f" % (rx, n, d_us, h_us), flush=True)
