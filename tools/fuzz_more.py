#!/usr/bin/env python3
"""One-off parity fuzz of the wider entry points on the GPU box, random regexes from the fixture
generator: (a) rj_match_all_batch vs the oracle per text, (b) sharded rj_scan_run with carried
selection state vs the documented-semantics oracle, (c) MatchFirst / MatchAnywhere vs MatchAll[0],
(d) texts of 40..200 KB vs the oracle.  usage: fuzz_more.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import rejit_amd
from checkers import Oracle
from make_golden import RegexGen, ALPHABETS

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
only = sys.argv[3] if len(sys.argv) > 3 else ""   # "shard": only part (b)
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 4711)
oracle = Oracle()
bad = {"batch": 0, "shard": 0, "first": 0, "big": 0}
done = {"batch": 0, "shard": 0, "first": 0, "big": 0}


def text_of(alphabet, n):
    return "".join(rng.choice(alphabet) for _ in range(n)).encode("latin1")


for i in range(cases):
    alphabet = rng.choice(ALPHABETS)
    rx = RegexGen(rng, alphabet).alt(2).encode("latin1")
    if isinstance(oracle.match_all(rx, b""), int):
        continue
    try:
        p = rejit_amd.Program(rx)
    except rejit_amd.RejitError:
        continue
    # (a) batch
    texts = [text_of(alphabet, rng.choice([0, 1, 3, 17, 64, 300, 1500, 2100])) for _ in range(rng.randrange(2, 12))]
    want = [oracle.match_all(rx, t) for t in texts] if only != "shard" else []
    got = p.match_all_batch(texts) if only != "shard" else want
    done["batch"] += 1
    if got != want:
        bad["batch"] += 1; print("BATCH", rx, [len(t) for t in texts])
    # (b) shards + carry (documented semantics)
    n = rng.choice([700, 2500, 6000])
    t = text_of(alphabet, n)
    spec = oracle.match_all_spec(rx, t)
    d = torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).cuda()
    sc = rejit_amd.Scan(p)
    cuts = sorted({0, n + 1, *[rng.randrange(0, n + 1) for _ in range(3)]})
    got, cur, prev_end, have = [], 0, 0, False
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        sc.run_tensor(d, own_begin=lo, own_end=hi, carry_cur=cur, carry_prev_end=prev_end, have_prev=have)
        part = sc.spans()
        got += part
        if part:
            b_, e_ = part[-1]
            cur, prev_end, have = (e_ if e_ > b_ else b_ + 1), e_, True
    done["shard"] += 1
    # (patterns at risk of the reference's ring artefact are replayed segment by segment, also in ranges: the
    # reference's own answer, which may differ from the documented semantics)
    if got != spec and not (p.info()["ring_artefact_risk"] and got == oracle.match_all(rx, t)):
        bad["shard"] += 1; print("SHARD", rx, n, cuts, p.info()["ring_artefact_risk"])
    if only == "shard":
        continue
    # (c) first / anywhere
    all_ = p.match_all(t)
    done["first"] += 1
    if p.match_first(t) != (all_[0] if all_ else None) or p.match_anywhere(t) != bool(all_):
        bad["first"] += 1; print("FIRST", rx, n)
    # (d) a bigger text now and then
    if i % 6 == 0:
        tb = text_of(alphabet, rng.choice([40000, 100000, 200000]))
        try:
            gotb = p.match_all(tb)
        except rejit_amd.RejitError as e:
            print("BIG: refused", rx, str(e)[:60]); continue
        done["big"] += 1
        if gotb != oracle.match_all(rx, tb):
            bad["big"] += 1; print("BIG", rx, len(tb))
print("done", done, "mismatches", bad)
