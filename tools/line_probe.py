"""Probe of the assertion-only path (emit_scan.hip): `^`, `$`, `^$` over a text with line breaks -- whole-call time,
kernel time, equality with the dense kernel (RJ_NO_EMIT=1 in a second process is the A/B).
    python tools/line_probe.py [bytes]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import rejit_amd
from rejit_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000_000
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
t = W.random_ascii_torch(n, 0xC0FFEE, dev)
nl = torch.arange(60, n, 61, device=dev)
t[nl] = 10
t[nl[::7] + 1] = 13          # some \r\n\r runs and empty lines
del nl
for rx in ("^", "$", "^$"):
    sc = rejit_amd.Scan(rejit_amd.Program(rx))
    t0 = time.perf_counter()
    k = sc.run(t.data_ptr(), n, stream=st)
    cold = time.perf_counter() - t0
    sc.run(t.data_ptr(), n, stream=st)
    wall, ms = [], []
    for _ in range(10):
        t0 = time.perf_counter()
        k = sc.run(t.data_ptr(), n, stream=st)
        wall.append(time.perf_counter() - t0)
        ms.append(sc.stats()["scan_ms"])
    sp = sc.spans_tensor(dev)
    d = W.span_digest_torch(sp)
    med = sorted(wall)[len(wall) // 2]
    print(f"{rx!r}: {k} matches, call median {med * 1e3:.3f} ms (min {min(wall) * 1e3:.3f}, first {cold * 1e3:.2f}), kernel {sum(ms) / len(ms):.3f} ms; "
          f"(n + 16 k) / t_call = {(n + 16 * k) / med / 1e9:.0f} GB/s = {(n + 16 * k) / med / 8e12:.3f} of HBM peak; digest {d['mix']:x} {d['sum_begin']:x}", flush=True)
