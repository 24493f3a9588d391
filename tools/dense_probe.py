#!/usr/bin/env python3
"""GPU: the dense patterns of the bench over random ASCII -- call and scan-kernel times, which path ran (bit streams /
scan_dense_walk), slow starts.  usage: dense_probe.py [bytes, default 1e9] [calls, default 6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rejit_amd
from rejit_amd import workloads as W

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
text = W.random_ascii_torch(n, 0xC0FFEE, dev)
st = torch.cuda.current_stream(dev).cuda_stream
patterns = (b"[a-f]+[0-9]", b"[~]", b"[@#]", b"[a-z]+", b"[0-9]+", b"[A-Z][a-z]+", b"[0-9][0-9][0-9]", b"^")
if os.environ.get("DENSE_PROBE_RX"):
    patterns = tuple(x.encode() for x in os.environ["DENSE_PROBE_RX"].split(" "))
for rx in patterns:
    p = rejit_amd.Program(rx)
    s = rejit_amd.Scan(p)
    times = []
    for k in range(calls):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        c = s.run(text.data_ptr(), n, stream=st)
        times.append((time.perf_counter() - t0) * 1e3)
    stt = s.stats()
    print(f"{rx.decode():18s} n={n:.2e} matches={c:10d} first {times[0]:8.3f} ms  best {min(times[1:]):8.3f} ms  scan kernel {stt['scan_ms']:7.3f} ms  "
          f"{n / min(times[1:]) / 1e9:7.1f} GB/s  stream_path={stt['stream_path']} slow_starts={stt['slow_starts']} retries={stt['retries']}", flush=True)
