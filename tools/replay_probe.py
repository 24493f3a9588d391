#!/usr/bin/env python3
"""GPU: the exact replay (exact_replay.hip) on ONE sync-free stretch: `.{0,2}.` over a text without line breaks -- every
byte keeps a thread alive, so the whole text is one segment (until round 4 replayed by one lane: 6 us per byte; now in
parts, exact_path = 2), checked against the oracle up to 16 MiB.  usage: replay_probe.py [MiB ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rejit_amd
from rejit_amd import workloads as W
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
for rx in (b".{0,2}.",):
    p = rejit_amd.Program(rx)
    print(rx, "ring_artefact_risk", p.info().get("ring_artefact_risk"))
    for mib in [int(a) for a in sys.argv[1:]] or [1, 4]:
        n = mib << 20
        t = W.random_ascii_torch(n, 7, dev, ord("a"), ord("e"))
        s = rejit_amd.Scan(p)
        t0 = time.perf_counter(); c = s.run(t.data_ptr(), n, stream=st); dt = time.perf_counter() - t0
        ok = ""
        if mib <= 16:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
            from checkers import Oracle
            ok = " == oracle: %s" % (Oracle().match_all(rx, t.cpu().numpy().tobytes()) == s.spans())
        print("  %4d MiB: %9d matches, %.3f s = %.3f us per byte, exact_path=%s%s" % (mib, c, dt, dt / n * 1e6, s.stats().get("exact_path"), ok), flush=True)
