import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch, rejit_amd
from rejit_amd import workloads as W
dev = torch.device("cuda:0")
t = W.fasta_stripped_torch(50_000_000, dev); n = int(t.numel())
for r in range(3):
    ms = rejit_amd.stream_read_probe(t.data_ptr(), n, 20)
    print("pattern", os.environ.get("RJ_PROBE_PATTERN", "0"), "500 MB: %.4f ms = %.1f GB/s" % (ms, n / ms / 1e6))
