#!/usr/bin/env python3
"""One-off extended parity fuzz on the GPU box: random regexes (the fixture generator) over texts of
1.5..9 KiB, so that the full-chunk fast paths (register pre-steps, pipelined scan loop, halos, wave
edges) are exercised, not only the guarded tail chunk.  usage: fuzz_large.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import rejit_amd
from checkers import Oracle
from make_golden import RegexGen, ALPHABETS

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
oracle = Oracle()
bad = checked = dense = skipped = 0
for i in range(cases):
    alphabet = rng.choice(ALPHABETS)
    rx = RegexGen(rng, alphabet).alt(2).encode("latin1")
    n = rng.choice([1500, 2047, 2048, 2053, 3100, 4096, 5000, 9000])
    text = "".join(rng.choice(alphabet) for _ in range(n)).encode("latin1")
    want = oracle.match_all(rx, text)
    if isinstance(want, int):
        skipped += 1
        continue
    try:
        p = rejit_amd.Program(rx)
        got = p.match_all(text)
    except rejit_amd.RejitError as e:
        print("ERROR", rx, e); bad += 1; continue
    dense += p.info()["scan_mode"] == 0
    checked += 1
    if got != want:
        bad += 1
        k = next((j for j in range(min(len(got), len(want))) if got[j] != want[j]), min(len(got), len(want)))
        print("MISMATCH", rx, n, "first diff at", k, got[k:k+2], want[k:k+2], "spec" if got == oracle.match_all_spec(rx, text) else "")
        if bad > 10: break
print(f"checked {checked} (dense {dense}, skipped {skipped}), mismatches {bad}")
