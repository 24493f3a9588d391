"""Probe of the general one-kernel count (plane_count.hip: GeneralShape) against the span pipeline of the same set, on random
ASCII with planted strings (bench.py's general_one_pass text) -- and of the headline set, the regression check for ExactShape.
    python tools/count_general_probe.py [bytes] [steps]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import rejit_amd
from rejit_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
t = W.random_ascii_torch(n, 5, dev)
rng = random.Random(3)
words12 = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(12)) for _ in range(9)]
words6 = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(6)) for _ in range(9)]
sets = {
    "general_one_pass (4 bases of 7)": (["alternation|strings", "prefix abcd|prefix 1234"], [b"alternation", b"strings", b"prefix abcd", b"prefix 1234"]),
    "nine 12-mers": ([w.decode() for w in words12], words12),
    "nine 6-mers": ([w.decode() for w in words6], words6),
    "one literal `regexp`": (["regexp"], [b"regexp"]),
    "one 12-mer": ([words12[0].decode()], [words12[0]]),
    "class windows (tolerance 1)": (["ab[cx]defgh", "zzz[0-9]yyyy"], [b"abcdefgh", b"zzz7yyyy"]),
}
probe = rejit_amd.stream_read_probe(t.data_ptr(), n, 10, st)
print("bytes %d; read-only ceiling %.3f ms = %.0f GB/s" % (n, probe, n / probe / 1e6))
for name, (rxs, strings) in sets.items():
    for k, o in enumerate(W.plant_offsets(n, 32, 400, seed=11)):
        W.plant(t, [o], strings[k % len(strings)])
    progs = [rejit_amd.Program(rx) for rx in rxs]
    row = []
    for counts_only in (True, False):
        if len(progs) == 1 and not counts_only:
            sc = rejit_amd.Scan(progs[0])
            run = lambda: [sc.run(t.data_ptr(), n, stream=st)]
            kern = lambda: sc.stats()["scan_ms"]
            how = lambda: "pipeline"
        elif len(progs) == 1:
            sc = rejit_amd.Scan(progs[0])
            run = lambda: [sc.count(t.data_ptr(), n, stream=st)]
            kern = lambda: sc.stats()["scan_ms"]
            how = lambda: "count_path=%d" % sc.stats()["count_path"]
        else:
            m = rejit_amd.MultiScan(progs)
            m.set_counts_only(counts_only)
            run = lambda: m.run(t.data_ptr(), n, stream=st)
            kern = lambda: m.scan_ms()
            how = lambda: "how=%d" % m.how
        c = run()
        for _ in range(3):
            run()
        torch.cuda.synchronize(dev)
        ks = []
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
            ks.append(kern())
        dt = (time.perf_counter() - t0) / steps
        row.append((c, how(), dt * 1e3, sum(ks) / len(ks)))
    (c1, h1, d1, k1), (c2, h2, d2, k2) = row
    print("%-34s counts %s %s | counts-only %s: call %.3f ms, kernel %.3f ms = %.0f GB/s = %.3f of peak, %.3f of ceiling | spans %s: call %.3f ms, scan kernel %.3f ms" % (
        name, c1, "==" if c1 == c2 else "!= %s" % c2, h1, d1, k1, n / k1 / 1e6, n / k1 / 1e6 / 8000, probe / k1, h2, d2, k2))
# the headline set on DNA
nf = n // 10
text = W.fasta_stripped_torch(nf, dev)
nn = int(text.numel())
progs = [rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS]
m = rejit_amd.MultiScan(progs)
m.set_counts_only(True)
c = m.run(text.data_ptr(), nn, stream=st)
ks = []
for _ in range(steps + 5):
    m.run(text.data_ptr(), nn, stream=st)
    ks.append(m.scan_ms())
ks = ks[5:]
probe = rejit_amd.stream_read_probe(text.data_ptr(), nn, 10, st)
print("regexdna nine, FASTA %d bytes: how=%d kernel %.4f ms (min %.4f) = %.3f of peak, %.3f of ceiling (%.4f ms)" % (nn, m.how, sum(ks) / len(ks), min(ks), nn / (sum(ks) / len(ks)) / 1e6 / 8000, probe / (sum(ks) / len(ks)), probe))
