#!/usr/bin/env python3
"""bench.py's jrep_10gb extra on its own (BASELINE configs[4] on one GPU: 100 000 files / 10 GB through rj_match_all_batch),
with RJ_COPY_THREADS sweeps run as child processes.   usage: jrep_probe.py [files] [bytes] [threads ...]"""
import os, subprocess, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
files = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
nbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000_000
sweep = sys.argv[3:]
if sweep and "RJ_COPY_THREADS" not in os.environ:
    for th in sweep:
        subprocess.call([sys.executable, __file__, str(files), str(nbytes)], env=dict(os.environ, RJ_COPY_THREADS=th))
    sys.exit(0)
import bench
args = types.SimpleNamespace(jrep_files=files, jrep_bytes=nbytes, no_cpu_baseline=True, jrep_threads=int(os.environ.get("JREP_THREADS", "1")))
out = {}
bench.jrep_extra(args, None, out)
r = out["jrep_10gb"]
print("JREP_THREADS=%s RJ_COPY_THREADS=%s: %s GB/s end to end, %s s, parity: %s" % (os.environ.get("JREP_THREADS", "1"), os.environ.get("RJ_COPY_THREADS", "default"), r["value"], r["seconds"], "parity_full_size" in r), flush=True)
