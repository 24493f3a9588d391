#!/usr/bin/env python3
"""The reference's regexdna.cc UNCHANGED on librejit_hip.so (oracle/_ref/regexdna_hip), wall time on the 50M-line input, with
RJ_TRACE_HOST=1 accounting of the library's host-text calls on stderr when the library was built with it.
usage: e2e_probe.py [fasta_n] [runs]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rejit_amd import workloads as W
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
raw = W.fasta_raw_torch(nf, torch.device("cuda:0")).cpu().numpy()
path = "/dev/shm/e2e_probe_%d.txt" % os.getpid()
raw.tofile(path)
del raw
try:
    for exe in ("regexdna_hip", "regexdna_ref"):
        p = os.path.join(ROOT, "oracle", "_ref", exe)
        if not os.path.exists(p):
            continue
        for _ in range(runs if exe == "regexdna_hip" else 1):
            with open(path, "rb") as fh:
                t0 = time.perf_counter()
                r = subprocess.run([p], stdin=fh, capture_output=True)
                dt = time.perf_counter() - t0
            print("%s: %.3f s rc=%d  %s" % (exe, dt, r.returncode, " ".join(r.stdout.decode().split()[1:18:2])), flush=True)
            if r.stderr:
                print(r.stderr.decode()[-600:])
    native = os.path.join(ROOT, "samples", "regexdna_gpu")
    if os.path.exists(native):
        with open(path, "rb") as fh:
            t0 = time.perf_counter(); r = subprocess.run([native], stdin=fh, capture_output=True); dt = time.perf_counter() - t0
        print("samples/regexdna_gpu: %.3f s" % dt)
finally:
    os.unlink(path)
