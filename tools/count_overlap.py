import os, sys, time
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, rejit_amd
from rejit_amd import workloads as W
steps = 200
dev = torch.device("cuda:0")
text = W.fasta_stripped_torch(50_000_000, dev); n = int(text.numel())
progs = [rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS]
for depth, two_streams in ((2, False), (2, True), (3, True), (4, True), (2, False)):
    streams = [torch.cuda.Stream(dev) for _ in range(depth)] if two_streams else [torch.cuda.current_stream(dev)] * depth
    ms = [rejit_amd.MultiScan(progs) for _ in range(depth)]
    for i, m in enumerate(ms):
        m.set_counts_only(True); m.set_timing(i == 0)
    busy = [False] * depth
    kt = []
    def loop(k, rec):
        for i in range(k):
            j = i % depth
            if busy[j]:
                ms[j].finish()
                if rec and j == 0: kt.append(ms[0].scan_ms())
            ms[j].start(text.data_ptr(), n, stream=streams[j].cuda_stream); busy[j] = True
        for d in range(depth):
            j = (k + d) % depth
            if busy[j]: ms[j].finish(); busy[j] = False
    loop(200, False); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(steps, True); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("depth %d streams %s: ms/step %.4f (%.3f of peak), kernel %.4f" % (depth, "own" if two_streams else "one", dt * 1e3, n / dt / 8e12, sum(kt) / len(kt)), flush=True)
