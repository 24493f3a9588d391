"""The steps-in-flight loop of the counts path only (for a kernel-trace timeline, or depth / step-count sweeps):
    python tools/count_flight.py [steps] [depth] [timed objects]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, rejit_amd
from rejit_amd import workloads as W
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_timed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
text = W.fasta_stripped_torch(50_000_000, dev); n = int(text.numel())
progs = [rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS]
ms = [rejit_amd.MultiScan(progs) for _ in range(depth)]
for i, m in enumerate(ms):
    m.set_counts_only(True); m.set_timing(i < n_timed)
busy = [False] * depth
def loop(k):
    for i in range(k):
        j = i % depth
        if busy[j]: ms[j].finish()
        ms[j].start(text.data_ptr(), n, stream=st); busy[j] = True
    for d in range(depth):
        j = (k + d) % depth
        if busy[j]: ms[j].finish(); busy[j] = False
loop(200); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); loop(steps); torch.cuda.synchronize()
    print("depth %d, %d steps, %d timed objects: ms/step %.4f" % (depth, steps, n_timed, (time.perf_counter() - t0) / steps * 1e3), flush=True)
