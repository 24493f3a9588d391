#!/usr/bin/env python3
"""GPU: synchronous regexdna steps (rj_multi_run, 500 MB) on a settled device: ms per call under the environment's
overrides (RJ_CLASSIFY_FORWARD, RJ_PLANE_LDS, ...); with rocprofv3 --kernel-trace around it the tail kernels' own durations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rejit_amd
from rejit_amd import workloads as W
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
progs = [rejit_amd.Program(p) for p in W.REGEXDNA_PATTERNS]
text = W.fasta_stripped_torch(50_000_000, dev); n = int(text.numel())
m = rejit_amd.MultiScan(progs)
for _ in range(300): m.run(text.data_ptr(), n, stream=st)
t0 = time.perf_counter()
for _ in range(200): c = m.run(text.data_ptr(), n, stream=st)
print("%.4f ms per synchronous step, scan %.4f, counts %d" % ((time.perf_counter() - t0) / 200 * 1e3, m.scan_ms(), sum(c)))
