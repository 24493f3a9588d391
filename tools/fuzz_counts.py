#!/usr/bin/env python3
"""Parity fuzz of MatchAllCount in one kernel (plane_count.hip / exact_count.h) on pattern sets that are NOT regexdna's:
random base windows of 8 bytes over random 4-letter alphabets (ASCII and bytes >= 0x80), one or two bases, 1..9 patterns
each an alternation of one or two "base with one position turned into a class (listed or negated) or another letter",
over random texts with planted strings one and two bytes off the bases; counts against the oracle, the span pipeline of
the same object beside them; how often the set took the one-kernel path is printed.
usage: fuzz_counts.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rejit_amd
from checkers import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
oracle = Oracle()
SAFE = [c for c in range(0x21, 0x7f) if chr(c) not in "\\[](){}|*+?.^$-"] + list(range(0xa1, 0xff))


def lit(c):
    return bytes([c])


bad = took = fell = refused = 0
for case in range(cases):
    # four letters the plane scan can tell apart: distinct 2-bit codes under some shift, distinct low nibbles
    while True:
        letters = rng.sample(SAFE, 4)
        if len({c & 15 for c in letters}) == 4 and any(len({(c >> sh) & 3 for c in letters}) == 4 for sh in range(7)):
            break
    nb = rng.choice([1, 2, 2])
    bases = []
    while len(bases) < nb:
        b = bytes(rng.choice(letters) for _ in range(8))
        if all(sum(x != y for x, y in zip(b, o)) >= 3 for o in bases):
            bases.append(b)
    patterns = []
    for _ in range(rng.randint(1, 9)):
        branches = []
        for _ in range(rng.choice([1, 2])):
            b = rng.choice(bases)
            j = rng.randrange(8)
            k = rng.random()
            if k < 0.15:
                mid = lit(b[j])                                        # the base itself
            elif k < 0.3:
                mid = lit(rng.choice(letters))                          # another letter
            elif k < 0.85:
                cls = rng.sample(letters, rng.randint(1, 3))
                mid = b"[" + b"".join(lit(c) for c in cls) + b"]"
            else:
                mid = b"[^" + lit(b[j]) + b"]"                         # anything but the base's letter (incl. every other byte)
            branches.append(b"".join(lit(c) for c in b[:j]) + mid + b"".join(lit(c) for c in b[j + 1:]))
        patterns.append(b"|".join(branches))
    n = rng.choice([5000, 33000, 70001, 200000, 400000])
    other = rng.sample(SAFE, 3)
    t = bytearray(rng.choices(letters + (other if rng.random() < 0.5 else []), k=n))
    for _ in range(n // 300):
        b = bytearray(rng.choice(bases))
        for _ in range(rng.choice([0, 1, 1, 1, 2])):
            b[rng.randrange(8)] = rng.choice(letters + other)
        at = rng.randrange(0, n - 8)
        t[at:at + 8] = b
    text = bytes(t)
    want = [len(oracle.match_all(rx, text)) for rx in patterns]
    try:
        progs = [rejit_amd.Program(rx) for rx in patterns]
        m = rejit_amd.MultiScan(progs)
        ok = m.set_counts_only(True)
        d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        got = m.run(d.data_ptr(), n)
        how = m.how
        m2 = rejit_amd.MultiScan(progs)
        got2 = m2.run(d.data_ptr(), n)
    except rejit_amd.RejitError as e:
        print("ERROR", patterns, e, flush=True)
        bad += 1
        continue
    took += how == 3
    fell += ok and how != 3
    refused += not ok
    if got != want or got2 != want:
        bad += 1
        print("MISMATCH", patterns, "n", n, "how", how, "counts-only", got, "spans", got2, "want", want, flush=True)
print("cases %d: mismatches %d; one-kernel path %d, voided -> span pipeline %d, shape refused %d" % (cases, bad, took, fell, refused))
