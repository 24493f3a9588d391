"""measurement: plane_count's kernel time under RJ_COUNT_DEBUG switches (each in a process of its own)"""
import os, subprocess, sys
code = r'''
import os, sys, time
sys.path.insert(0, ".")
import torch, rejit_amd
from rejit_amd import workloads as W
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
text = W.fasta_stripped_torch(50_000_000, dev); n = int(text.numel())
m = rejit_amd.MultiScan([rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS]); m.set_counts_only(True)
ms = []
import time
for i in range(30):
    try:
        m.run(text.data_ptr(), n, stream=st)
    except Exception as e:
        pass
    ms.append(m.scan_ms())
ms = ms[5:]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    try:
        m.run(text.data_ptr(), n, stream=st)
    except Exception as e:
        pass
torch.cuda.synchronize()
step = (time.perf_counter() - t0) / 100
print("sync step %.4f ms" % (step * 1e3), end="  ")
print(os.environ.get("RJ_COUNT_DEBUG", "0"), os.environ.get("RJ_SCAN_GRID", "-"), "kernel %.4f ms (min %.4f) how=%d" % (sum(ms)/len(ms), min(ms), m.how), flush=True)
'''
for dbg, chunks in [("0", ""), ("512", ""), ("128", ""), ("640", ""), ("448", "")]:
    env = dict(os.environ, RJ_COUNT_DEBUG=dbg)
    if chunks:
        env["RJ_SCAN_GRID"] = chunks
    subprocess.run([sys.executable, "-c", code], env=env)
