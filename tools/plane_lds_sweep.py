#!/usr/bin/env python3
"""GPU: plane_scan's duration against its residency (RJ_PLANE_LDS = unused LDS per workgroup; 160 KiB per CU), each value in a
process of its own (the override is read once): synchronous rj_multi_run calls, the scan kernel's own time."""
import os, subprocess, sys
child = r'''
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1]))))
import torch, rejit_amd
from rejit_amd import workloads as W
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
progs = [rejit_amd.Program(p) for p in W.REGEXDNA_PATTERNS]
for nf in (50_000_000, 250_000_000):
    text = W.fasta_stripped_torch(nf, dev); n = int(text.numel())
    m = rejit_amd.MultiScan(progs)
    for _ in range(3): m.run(text.data_ptr(), n, stream=st)
    ms = []
    for _ in range(15):
        m.run(text.data_ptr(), n, stream=st); ms.append(m.scan_ms())
    ms.sort()
    print("   n=%d: scan kernel median %.4f ms min %.4f  -> %.3f of HBM peak" % (n, ms[len(ms)//2], ms[0], n / ms[len(ms)//2] / 1e6 / 8000), flush=True)
    del text, m
'''
for lds in [0, 23000, 27000, 32000, 40000, 54000]:
    env = dict(os.environ, RJ_PLANE_LDS=str(lds))
    print("RJ_PLANE_LDS=%d (%s workgroups per CU)" % (lds, "7 (registers)" if lds < 23000 else str(163840 // lds)), flush=True)
    subprocess.run([sys.executable, "-c", child, os.path.abspath(__file__)], env=env)
