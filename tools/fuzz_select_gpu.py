#!/usr/bin/env python3
"""GPU fuzz of dense mode through the C ABI with the generator of tools/fuzz_select.py (chain patterns whose candidates can
overlap, among others): rj_scan_run vs the oracle on texts from packed to sparse, 20 KB .. 1 MiB; how many runs the bit-stream
kernel answered (stream_path), how many it declared void (repeated on scan_dense_walk).
usage: fuzz_select_gpu.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rejit_amd
from checkers import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
oracle = Oracle()
LETTERS = "abcdefgh0123"


def cls():
    k = rng.random()
    if k < 0.35:
        return rng.choice(LETTERS)
    if k < 0.5:
        return "."
    members = rng.sample(LETTERS, rng.randint(2, 5))
    return ("[^" if rng.random() < 0.15 else "[") + "".join(members) + "]"


def word():
    return "".join(cls() for _ in range(rng.randint(1, 4)))


def pattern():
    k = rng.random()
    if k < 0.5:
        p = word()
    elif k < 0.8:
        p = word() + "|" + word()
    else:
        p = word() + cls() + "{%d,%d}" % tuple(sorted((rng.randint(1, 3), rng.randint(1, 4))))
    return p.encode()


bad = streams = other = 0
for case in range(cases):
    rx = pattern()
    alphabet = rng.choice([LETTERS, LETTERS[:4], LETTERS + "xyzwvu   \n", LETTERS + "".join(chr(c) for c in range(0x40, 0x7f))])
    n = rng.choice([20000, 33000, 40000, 70000, 100000, 300000, 1 << 20])
    piece = "".join(rng.choice(alphabet) for _ in range(min(n, 150000)))
    text = (piece * (n // len(piece) + 1))[:n].encode()
    want = oracle.match_all(rx, text)
    if isinstance(want, int):
        continue
    try:
        sc = rejit_amd.Scan(rejit_amd.Program(rx))
        d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        sc.run_tensor(d)
        got = sc.spans()
        if sc.stats()["stream_path"]:
            streams += 1
        else:
            other += 1
        # the same object again (a void run switched the stream kernel off; a good one left hints)
        sc.run_tensor(d)
        again = sc.spans()
    except rejit_amd.RejitError as e:
        got = again = ("ERROR", str(e))
    if got != want or again != want:
        bad += 1
        print("MISMATCH", rx, n, len(alphabet), len(want), len(got) if isinstance(got, list) else got, flush=True)
print("cases %d: bit-stream kernel %d, other paths %d, mismatches %d" % (cases, streams, other, bad))
