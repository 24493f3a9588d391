"""Probe of the bit-plane multi-pattern scan (plane_scan.hip): the nine regexdna counts in one pass over a
stripped FASTA text -- kernel time, step time, equality with one scan kernel per pattern.
    python tools/plane_probe.py [fasta_n] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import rejit_amd
from rejit_amd import workloads as W

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
text = W.fasta_stripped_torch(nf, dev)
n = int(text.numel())
progs = [rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS]


def run(mode, label):
    m = rejit_amd.MultiScan(progs)
    m.set_mode(mode)
    t0 = time.perf_counter()
    counts = m.run(text.data_ptr(), n, stream=st)
    cold = time.perf_counter() - t0
    m.run(text.data_ptr(), n, stream=st)
    torch.cuda.synchronize(dev)
    ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        m.run(text.data_ptr(), n, stream=st)
        ms.append(m.scan_ms())
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    k = sum(ms) / len(ms)
    print(f"{label}: step {dt * 1e3:.4f} ms, scan kernel(s) {k:.4f} ms (min {min(ms):.4f}), n/t_kernel {n / k / 1e6:.1f} GB/s "
          f"= {n / k / 1e6 / 8000:.3f} of HBM peak, first call {cold * 1e3:.2f} ms, how={m.how}", flush=True)
    spans = [m.scan(i).spans() for i in range(len(progs))] if nf <= 5_000_000 else None
    return counts, spans


c0, s0 = run(0, "mode 0 (one pass, plane scan)")
c3, s3 = run(3, "mode 3 (one kernel per pattern)")
print("counts", c0)
assert c0 == c3, (c0, c3)
assert s0 == s3
