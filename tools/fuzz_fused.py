#!/usr/bin/env python3
"""One-off: random DNA-style pattern sets through rj_multi (fused scan) vs single runs."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, rejit_amd
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream

def word(k):
    out = ""
    for _ in range(k):
        if rng.random() < 0.2:
            out += "[" + "".join(sorted(rng.sample("acgt", rng.choice([2, 3])))) + "]"
        else:
            out += rng.choice("acgt")
    return out

bad = fused_runs = 0
for trial in range(60):
    P = rng.randrange(2, 13)
    pats = [(word(rng.choice([6, 7, 8, 9])) + ("|" + word(rng.choice([6, 7, 8])) if rng.random() < 0.7 else "")).encode() for _ in range(P)]
    n = rng.choice([100, 5000, 300000, 3000000])
    t = np.frombuffer(bytes(rng.choice(b"acgt") for _ in range(min(n, 300000))) * (max(1, n // 300000)), dtype=np.uint8).copy()
    d = torch.from_numpy(t).to(dev); n = int(d.numel())
    progs = [rejit_amd.Program(p) for p in pats]
    multi = rejit_amd.MultiScan(progs)
    counts = multi.run(d.data_ptr(), n, stream=st)
    fused_runs += multi.fused
    for i, p in enumerate(progs):
        sc = rejit_amd.Scan(p); c = sc.run(d.data_ptr(), n, stream=st)
        if c != counts[i] or sc.spans() != multi.scan(i).spans():
            bad += 1; print("MISMATCH", pats[i], n, c, counts[i], "fused" if multi.fused else "seq")
print("trials 60, fused", fused_runs, "mismatches", bad)
