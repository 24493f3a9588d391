"""two-in-flight loop of the counts path only (for a kernel-trace timeline): python tools/count_flight.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, rejit_amd
from rejit_amd import workloads as W
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
text = W.fasta_stripped_torch(50_000_000, dev); n = int(text.numel())
progs = [rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS]
ms = [rejit_amd.MultiScan(progs) for _ in range(2)]
for i, m in enumerate(ms):
    m.set_counts_only(True); m.set_timing(i == 0)
busy = [False, False]
def loop(k):
    for i in range(k):
        j = i % 2
        if busy[j]: ms[j].finish()
        ms[j].start(text.data_ptr(), n, stream=st); busy[j] = True
    for j in (k % 2, (k + 1) % 2):
        if busy[j]: ms[j].finish(); busy[j] = False
loop(steps); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(steps); torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / steps * 1e3)
