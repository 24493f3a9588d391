#!/usr/bin/env python3
"""Parity + DETERMINISM fuzz under memory churn (round 5): random patterns (the fixture generator, ASCII and high-byte
alphabets) over texts of 33 KiB .. 600 KiB; every (pattern, text) runs several times on a FRESH rj_scan each, with device
allocations of other sizes coming and going in between (they change where buffers land and how launches are timed), and
must give the oracle's answer every time.  A result that differs from run to run is a race or a read of memory nobody
wrote -- this is how the untagged granule of offsets_gather_check was found (54 bad runs of 300).
usage: fuzz_churn.py [cases] [seed] [runs per case]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import rejit_amd
from checkers import Oracle
from make_golden import RegexGen, ALPHABETS, ALPHABETS_HI

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
oracle = Oracle()
churn = random.Random(7)
bad = flaky = checked = skipped = 0
paths = {}
for i in range(cases):
    alphabet = rng.choice(ALPHABETS if i % 3 else [a.replace("\x00", "") for a in ALPHABETS_HI])
    rx = RegexGen(rng, alphabet).alt(2).encode("latin1")
    n = rng.choice([33000, 40000, 65536, 70001, 131072, 150000, 300000, 600000])
    text = "".join(rng.choice(alphabet) for _ in range(n)).encode("latin1")
    want = oracle.match_all(rx, text)
    if isinstance(want, int):
        skipped += 1
        continue
    d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    results = []
    for r in range(runs):
        junk = [torch.full((churn.choice([1 << 12, 1 << 16, 1 << 20, 4 << 20]),), churn.randrange(256), dtype=torch.uint8, device="cuda")
                for _ in range(churn.randrange(0, 5))]
        del junk
        try:
            p = rejit_amd.Program(rx)
            sc = rejit_amd.Scan(p)
            sc.run_tensor(d)
            results.append(sc.spans())
            st = sc.stats()
            key = ("linear" if st["linear_path"] else "exact" if st["exact_path"] else "stream" if st["stream_path"] else
                   "large" if st["large_path"] else "dense" if p.info()["scan_mode"] == 0 else "windows")
            paths[key] = paths.get(key, 0) + 1
        except rejit_amd.RejitError as e:
            results.append(("ERROR", str(e)))
    checked += 1
    if any(r != results[0] for r in results):
        flaky += 1
        print("FLAKY", rx, n, [len(r) for r in results], flush=True)
    if any(r != want for r in results):
        bad += 1
        print("MISMATCH", rx, n, "want", len(want), [len(r) if isinstance(r, list) else r for r in results], flush=True)
print("cases %d checked %d skipped %d x %d runs: mismatches %d, run-to-run differences %d; paths %s" % (cases, checked, skipped, runs, bad, flaky, paths))
