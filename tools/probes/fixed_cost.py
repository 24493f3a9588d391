"""What part of a scan launch does not scale with the text?  Kernel duration (HIP events of the launch itself) of
plane_count<ExactShape<2>> (counts only), plane_count<ListShape<2>> (span lists) and the read-only probe over prefixes of the
stripped FASTA text, 1 MB ... 500 MB: T(n) = a + n / b.
    python tools/probes/fixed_cost.py [fasta_n] [launches]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import rejit_amd
from rejit_amd import workloads as W

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
text = W.fasta_stripped_torch(nf, dev)
n_all = int(text.numel())
progs = [rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS]
sizes = [s for s in (1 << 20, 4 << 20, 16 << 20, 64 << 20, 128 << 20, 256 << 20) if s < n_all] + [n_all]


def kernel_ms(counts_only, n):
    m = rejit_amd.MultiScan(progs)
    m.set_counts_only(counts_only)
    for _ in range(5):
        m.run(text.data_ptr(), n, stream=st)
    ms = []
    for _ in range(launches):
        m.run(text.data_ptr(), n, stream=st)
        ms.append(m.scan_ms())
    ms.sort()
    return ms[len(ms) // 2], ms[0]


for r in range(2):
    for n in sizes:
        ce, cemin = kernel_ms(True, n)
        le, lemin = kernel_ms(False, n)
        pr = rejit_amd.stream_read_probe(text.data_ptr(), n, 10, st)
        print(f"round {r} n={n:>10}: counts kernel {ce * 1e3:8.2f} us (min {cemin * 1e3:8.2f})  list kernel {le * 1e3:8.2f} us (min {lemin * 1e3:8.2f})  "
              f"read probe {pr * 1e3:8.2f} us   ideal at 7 TB/s {n / 7e6:8.2f} us", flush=True)
