#!/usr/bin/env python3
"""Parity fuzz of REUSED objects under memory churn (round 5): what fuzz_churn.py leaves out.  One rj_scan per pattern and
one rj_multi per set live across many texts of different sizes (33 KiB .. 3 MiB, so single- and multi-tile geometries and
the carry scan all come up on the same object), device allocations come and go in between, and every run must give the
oracle's spans.  A difference here is state a run left behind for the next one (epochs, tickets, high-water marks).
usage: fuzz_reuse.py [sets] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import rejit_amd
from checkers import Oracle
from make_golden import RegexGen, ALPHABETS, ALPHABETS_HI

sets = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
oracle = Oracle()
churn = random.Random(11)
st = torch.cuda.current_stream().cuda_stream
SIZES = [33000, 65536, 70001, 300000, 1 << 20, (1 << 20) + 77, 3000000]


def dna_word(k):
    out = ""
    for _ in range(k):
        out += ("[" + "".join(sorted(rng.sample("acgt", rng.choice([2, 3])))) + "]") if rng.random() < 0.2 else rng.choice("acgt")
    return out


def make_set(kind, alphabet):
    if kind == "dna":
        return [(dna_word(rng.choice([6, 7, 8, 9])) + ("|" + dna_word(rng.choice([6, 7, 8])) if rng.random() < 0.7 else "")).encode()
                for _ in range(rng.randrange(2, 10))]
    return [RegexGen(rng, alphabet).alt(2).encode("latin1") for _ in range(rng.randrange(2, 6))]


bad = runs = 0
kinds = {}
for s in range(sets):
    kind = "dna" if s % 2 == 0 else "general"
    alphabet = "acgt" if kind == "dna" else rng.choice(ALPHABETS if s % 4 == 1 else [a.replace("\x00", "") for a in ALPHABETS_HI])
    pats = make_set(kind, alphabet)
    try:
        progs = [rejit_amd.Program(p) for p in pats]
    except rejit_amd.RejitError:
        continue
    multi = rejit_amd.MultiScan(progs)
    singles = [rejit_amd.Scan(p) for p in progs]
    for t in range(5):
        n = rng.choice(SIZES)
        piece = "".join(rng.choice(alphabet) for _ in range(min(n, 200000)))
        text = (piece * (n // len(piece) + 1))[:n].encode("latin1")
        d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        junk = [torch.full((churn.choice([1 << 12, 1 << 16, 1 << 20, 4 << 20]),), churn.randrange(256), dtype=torch.uint8, device="cuda")
                for _ in range(churn.randrange(0, 5))]
        del junk
        want = [oracle.match_all(p, text) for p in pats]
        if any(isinstance(w, int) for w in want):
            continue
        try:
            counts = multi.run(d.data_ptr(), n, stream=st)
            got_multi = [multi.scan(i).spans() for i in range(len(pats))]
        except rejit_amd.RejitError as e:
            counts, got_multi = None, [("ERROR", str(e))] * len(pats)
        key = kind + ("/fused" if multi.fused else "/seq")
        kinds[key] = kinds.get(key, 0) + 1
        for i, p in enumerate(pats):
            try:
                singles[i].run(d.data_ptr(), n, stream=st)
                got_one = singles[i].spans()
            except rejit_amd.RejitError as e:
                got_one = ("ERROR", str(e))
            runs += 1
            if got_one != want[i]:
                bad += 1
                print("MISMATCH single", p, n, len(want[i]), len(got_one) if isinstance(got_one, list) else got_one, flush=True)
            if got_multi[i] != want[i] or (counts is not None and counts[i] != len(want[i])):
                bad += 1
                print("MISMATCH multi", p, n, len(want[i]), len(got_multi[i]) if isinstance(got_multi[i], list) else got_multi[i], flush=True)
print("sets %d, pattern runs %d on reused objects: mismatches %d; %s" % (sets, runs, bad, kinds))
