#!/usr/bin/env python3
"""Parity fuzz of MatchAllCount in one kernel for the GENERAL shapes (plane_count.hip: GeneralShape; round 6): sets of 1..9
patterns, each an alternation of 1..3 literals of 4..16 bytes -- some with one position turned into a class, some sharing
prefixes (`abcd|abcdefgh`: the longest wins) -- over a 2..6-letter alphabet (dense: matches, pairs and chains of overlapping
matches are common) or over random ASCII with planted strings (sparse); whole texts and own ranges.  Counts against the
oracle, counts + first / last matches against the span pipeline of the same set, the single-pattern entry points
(rj_scan_count) beside them; how often the set took the one-kernel path is printed.
usage: fuzz_counts_general.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rejit_amd
from checkers import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
oracle = Oracle()
LETTERS = b"abcdefghijklmnopqrstuvwxyz0123456789"
ASCII = bytes(range(ord("0"), ord("z")))

bad = took = fell = refused = singles = 0
for case in range(cases):
    dense = rng.random() < 0.5
    alphabet = bytes(rng.sample(list(LETTERS), rng.randint(2, 6))) if dense else LETTERS
    lits = []

    def literal():
        k = rng.choice([4, 5, 6, 6, 7, 8, 8, 9, 11, 12, 16]) if rng.random() < 0.2 else rng.choice([6, 7, 8, 8, 9, 10, 12, 14, 16])
        if lits and rng.random() < 0.25:                     # share a prefix with an earlier one
            base = rng.choice(lits)
            s = (base + bytes(rng.choice(alphabet) for _ in range(16)))[:k]
        else:
            s = bytes(rng.choice(alphabet) for _ in range(k))
        lits.append(s)
        return s

    patterns = []
    for _ in range(rng.choice([1, 1, 2, 2, 3, 4, 6, 9])):
        branches = []
        for _ in range(rng.choice([1, 1, 2, 3])):
            s = literal()
            if rng.random() < 0.25:                          # one class position
                j = rng.randrange(len(s))
                cls = bytes(set(rng.choice(alphabet) for _ in range(rng.randint(1, 3))) | {s[j]})
                neg = rng.random() < 0.2
                branches.append(s[:j] + (b"[^" if neg else b"[") + cls + b"]" + s[j + 1:])
            else:
                branches.append(s)
        patterns.append(b"|".join(branches))
    n = rng.choice([16, 17, 24, 40, 100, 2047, 2048, 2049, 2056, 3000, 4095, 4097, 33000, 70001, 200000])
    if dense:
        t = bytearray(rng.choices(alphabet, k=n))
    else:
        t = bytearray(rng.choices(ASCII, k=n))
    for _ in range(n // 200 + 2):
        s = bytearray(rng.choice(lits))
        if len(s) > n:
            continue
        if rng.random() < 0.3:
            s[rng.randrange(len(s))] = rng.choice(alphabet)
        at = rng.choice([0, n - len(s), rng.randrange(0, n - len(s) + 1)])     # (also at the very begin and end of the text)
        t[at:at + len(s)] = s
    text = bytes(t)
    own = None if rng.random() < 0.7 else tuple(sorted((rng.randrange(0, n + 1), rng.randrange(0, n + 1))))
    if own is not None:
        # an independent range: the selection starts afresh at own_begin (include/rejit_hip.h: no state is carried in)
        spans = [[(b + own[0], e + own[0]) for b, e in oracle.match_all(rx, text[own[0]:]) if b + own[0] < own[1]] for rx in patterns]
    else:
        spans = [oracle.match_all(rx, text) for rx in patterns]
    want = [len(sp) for sp in spans]
    want_bounds = [None if not sp else (sp[0][0], sp[0][1], sp[-1][0], sp[-1][1]) for sp in spans]
    kw = {} if own is None else {"own_begin": own[0], "own_end": own[1]}
    try:
        progs = [rejit_amd.Program(rx) for rx in patterns]
        m = rejit_amd.MultiScan(progs)
        ok = m.set_counts_only(True)
        d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        got = m.run(d.data_ptr(), n, **kw)
        how = m.how
        gb = m.bounds()
        one = None
        if own is None and rng.random() < 0.3:
            i = rng.randrange(len(progs))
            sc = rejit_amd.Scan(progs[i])
            one = (i, sc.count(d.data_ptr(), n), sc.stats()["count_path"])
            singles += one[2]
    except rejit_amd.RejitError as e:
        print("ERROR", patterns, e, flush=True)
        bad += 1
        continue
    took += how == 3
    fell += ok and how != 3
    refused += not ok
    if got != want or gb != want_bounds or (one is not None and one[1] != want[one[0]]):
        bad += 1
        print("MISMATCH", patterns, "n", n, "own", own, "how", how, "counts", got, "want", want, "bounds", gb, "want", want_bounds, "single", one, flush=True)
print("cases %d: mismatches %d; one-kernel path %d, voided -> span pipeline %d, shape refused %d; single-pattern kernel runs %d" % (cases, bad, took, fell, refused, singles))
