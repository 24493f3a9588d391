#!/usr/bin/env python3
"""GPU fuzz of the PAIR kernels (run_scan.hip: pair_summary / pair_resolve / pair_emit) against the oracle: random `Q L* Q` patterns (one or
two Q bytes, L a class or a complement, resets of every density), texts of 1 B .. 400 KB whole, as counts and in own ranges.  Run with
RJ_NO_SMALL=1 to send the short texts through the kernels too and RJ_RUN_TWO_LEVELS=<tiles> for the two-level resolve.
usage: fuzz_pairs.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rejit_amd
from checkers import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
oracle = Oracle()
pool = list(b"abxy01\"'%|\n ,") + [0x80, 0xfe]


def esc(c):   # (the reference's dialect has no escapes inside brackets: the byte itself; the pool holds none of `\\ [ ] ^ -`)
    return bytes([c])


bad = took = other = 0
for case in range(cases):
    alphabet = rng.sample(pool, rng.randint(2, 8))
    q = sorted(set(rng.sample(alphabet, rng.randint(1, 2))))
    rest = [c for c in alphabet if c not in q]
    qcls = b"[" + b"".join(esc(c) for c in q) + b"]"
    if rng.random() < 0.5 or not rest:
        resets = sorted(set(rng.sample(rest, rng.randint(0, min(2, len(rest)))))) if rest else []
        lcls = b"[^" + b"".join(esc(c) for c in q + resets) + b"]"
    else:
        lcls = b"[" + b"".join(esc(c) for c in sorted(set(rng.sample(rest, rng.randint(1, len(rest)))))) + b"]"
    rx = qcls + lcls + b"*" + qcls
    n = rng.choice([1, 2, 31, 33, 700, 2047, 2049, 8191, 8193, 17000, 40000, 100000, 400000])
    mode = rng.random()
    if mode < 0.4:
        text = bytes(rng.choices(alphabet, k=n))
    else:
        text = bytearray(rng.choices(rest[:2] if rest else alphabet, k=n))
        for _ in range(rng.choice([0, 1, 2, 5, 40, 400, n // 50])):
            text[rng.randrange(n)] = rng.choice(alphabet)
        text = bytes(text)
    want = oracle.match_all(rx, text)
    if isinstance(want, int):
        continue
    try:
        sc = rejit_amd.Scan(rejit_amd.Program(rx))
        d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        kind = rng.random()
        if kind < 0.6:
            k = sc.run(d.data_ptr(), n)
            got = sc.spans()
        elif kind < 0.8:
            k = sc.count(d.data_ptr(), n)
            got = want
        else:
            ob = rng.randrange(n)
            oe = rng.randrange(ob, n + 2)
            k = sc.run(d.data_ptr(), n, own_begin=ob, own_end=oe)
            got = sc.spans()
            alt = [(b + ob, e + ob) for b, e in oracle.match_all(rx, text[ob:]) if b + ob < oe]
            want = alt if got == alt else [m for m in want if ob <= m[0] < oe]
        st = sc.stats()
    except rejit_amd.RejitError as e:
        print("ERROR", rx, e, flush=True)
        bad += 1
        continue
    took += 1 if st["run_path"] == 2 else 0
    other += 0 if st["run_path"] == 2 else 1
    if got != want or k != len(want):
        bad += 1
        print("MISMATCH", rx, "n", n, "run_path", st["run_path"], "got", k, got[:3], "want", len(want), want[:3], flush=True)
print("cases %d: mismatches %d; pair kernels %d, other paths %d" % (cases, bad, took, other))
