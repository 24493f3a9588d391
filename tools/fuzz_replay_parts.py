#!/usr/bin/env python3
"""GPU parity fuzz of the exact replay IN PARTS (exact_replay.hip, round 4): random regexes (the fixture generator) that are
at risk of the reference's ring artefact, over texts of 150..400 KB without the bytes that would give synchronisation points
every few bytes (no line breaks; alphabets of the pattern's own letters), so that segments are longer than 64 KiB and go through
xr_round / xr_walk / xr_emit / xr_join -- against the oracle.  usage: fuzz_replay_parts.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import rejit_amd
from checkers import Oracle
from make_golden import RegexGen, ALPHABETS

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
oracle = Oracle()
dev = torch.device("cuda:0")
bad = checked = parts = skipped = 0
tried = 0
while checked < cases and tried < cases * 40:
    tried += 1
    alphabet = rng.choice(ALPHABETS).replace("\n", "").replace("\r", "")
    if len(alphabet) < 2:
        continue
    rx = RegexGen(rng, alphabet).alt(2).encode("latin1")
    try:
        p = rejit_amd.Program(rx)
    except rejit_amd.RejitError:
        continue
    if not p.info()["ring_artefact_risk"]:
        continue
    letters = [c for c in alphabet if c.isalnum()] or list(alphabet)
    n = rng.choice([150_000, 200_001, 400_000])
    text = "".join(rng.choice(letters[: rng.choice([1, 2, 3, len(letters)])]) for _ in range(n)).encode("latin1")
    want = oracle.match_all(rx, text)
    if isinstance(want, int):
        skipped += 1
        continue
    t = torch.from_numpy(np.frombuffer(text + b"\0" * 16, dtype=np.uint8).copy()).to(dev)
    s = rejit_amd.Scan(p)
    try:
        c = s.run(t.data_ptr(), n)
        got = s.spans()
    except rejit_amd.RejitError as e:
        print("ERROR", rx, e); bad += 1; continue
    st = s.stats()
    checked += 1
    parts += st["exact_path"] == 2
    if got != want:
        bad += 1
        k = next((j for j in range(min(len(got), len(want))) if got[j] != want[j]), min(len(got), len(want)))
        print("MISMATCH", rx, n, "exact_path", st["exact_path"], "first diff at", k, got[k:k + 2], want[k:k + 2],
              "== documented semantics" if got == oracle.match_all_spec(rx, text) else "")
        if bad > 10:
            break
print(f"checked {checked} at-risk patterns ({parts} through the parts, {skipped} skipped, {tried} tried), mismatches {bad}")
