import os, sys
sys.path.insert(0, "/root/repo")
import torch, rejit_amd
from rejit_amd import workloads as W
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
t = W.fasta_stripped_torch(50_000_000, dev)
m = rejit_amd.MultiScan([rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS])
for _ in range(5): m.run(t.data_ptr(), t.numel(), stream=st)
ms = []
for _ in range(30):
    m.run(t.data_ptr(), t.numel(), stream=st); ms.append(m.scan_ms())
ms.sort()
print(os.environ.get("RJ_PLANE_CHUNKS"), "plane_scan median %.4f min %.4f ms" % (ms[15], ms[0]))
