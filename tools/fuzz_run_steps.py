#!/usr/bin/env python3
"""Parity fuzz of the run form of dense_streams' steps (dense_streams.h: rj_stream_runs, round 6): `X+` and `X+ Y` patterns with
random classes over texts whose runs are short, of 16..47 bytes (decided by the run form, undecided by the sixteen generic
steps) and longer than the lane's window (the scalar walk / the run kernels), whole texts and own ranges; against the oracle.
usage: fuzz_run_steps.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rejit_amd
from checkers import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
oracle = Oracle()
bad = stream = runk = other = 0
for case in range(cases):
    letters = rng.sample(list(b"abcdefghijklmnop0123456789 .,\x80\xe9"), rng.randint(3, 8))
    k = rng.randint(1, max(1, len(letters) - 2))
    X, rest = letters[:k], letters[k:]
    esc = lambda c: b"\\x%02x" % c if c >= 0x7f or chr(c) in "\\[]^-." else bytes([c])
    xs = b"[" + b"".join(esc(c) for c in X) + b"]"
    if rng.random() < 0.5:
        rx = xs + b"+"
    else:
        Y = rng.sample(rest, rng.randint(1, len(rest)))
        rx = xs + b"+[" + b"".join(esc(c) for c in Y) + b"]"
    n = rng.choice([5000, 40000, 70001, 200000])
    p_x = rng.choice([0.3, 0.7, 0.9, 0.97, 0.995])      # mean run lengths from 1.4 to 200 bytes
    t = bytearray(n)
    for i in range(n):
        t[i] = rng.choice(X) if rng.random() < p_x else rng.choice(rest)
    text = bytes(t)
    kw = {}
    want = oracle.match_all(rx, text)
    try:
        prog = rejit_amd.Program(rx)
        sc = rejit_amd.Scan(prog)
        d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        kcount = sc.run(d.data_ptr(), n, **kw)
        got = sc.spans()
        st = sc.stats()
    except rejit_amd.RejitError as e:
        print("ERROR", rx, e, flush=True)
        bad += 1
        continue
    stream += st["stream_path"]
    runk += st["run_path"]
    other += 1 - st["stream_path"] - st["run_path"]
    if got != want or kcount != len(want):
        bad += 1
        print("MISMATCH", rx, "n", n, "p_x", p_x, st, "got", len(got), got[:3], "want", len(want), want[:3], flush=True)
print("cases %d: mismatches %d; dense_streams %d, run kernels %d, other paths %d" % (cases, bad, stream, runk, other))
