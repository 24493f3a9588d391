#!/usr/bin/env python3
"""Parity fuzz of the run kernels (run_scan.hip) with RJ_RUNS_FIRST=1 -- every pattern of the four shapes goes there at once,
whatever its text: random classes (listed, ranges, negated, bytes >= 0x80) for A, L and B, texts of every break density and
sizes around the kernels' units, whole texts / independent ranges / ranges under a carried-in match; against the oracle.
usage: fuzz_runs.py [cases] [seed]"""
import os, random, sys
os.environ["RJ_RUNS_FIRST"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rejit_amd
from checkers import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
oracle = Oracle()
POOL = list(b"abcdxyz019<>\"\n ") + [0x80, 0xa5, 0xff]


def cls(alphabet):
    k = rng.random()
    members = sorted(set(rng.sample(alphabet, rng.randint(1, max(1, min(4, len(alphabet)))))))
    def esc(c):
        return b"\\x%02x" % c if (c >= 0x7f or c < 0x20 or chr(c) in "\\[]^-") else bytes([c])
    if k < 0.25 and len(members) == 1 and chr(members[0]).isalnum():
        return bytes([members[0]]), set(members)
    if k < 0.8:
        return b"[" + b"".join(esc(c) for c in members) + b"]", set(members)
    return b"[^" + b"".join(esc(c) for c in members) + b"]", set(range(256)) - set(members)


bad = took = refused = 0
for case in range(cases):
    alphabet = rng.sample(POOL, rng.randint(2, 7))
    shape = rng.choice(["X+", "AL*", "AL*B", "X+B", "A.*B", "AL+", "AL+B", "A.+B"])
    a, aset = cls(alphabet)
    l, lset = cls(alphabet)
    b, bset = cls(alphabet)
    if shape == "X+":
        rx = a + b"+"
    elif shape == "AL*":
        rx = a + l + b"*"
    elif shape == "AL*B":
        rx = a + l + b"*" + b
    elif shape == "X+B":
        rx = a + b"+" + b
    elif shape == "AL+":
        rx = a + l + b"+"
    elif shape == "AL+B":
        rx = a + l + b"+" + b
    elif shape == "A.+B":
        rx = a + b".+" + b
    else:
        rx = a + b".*" + b
    if rng.random() < 0.25:   # `^` / `$` around the shape (RunPlan::bol / eol)
        rx = rng.choice([b"^" + rx, rx + b"$", b"^" + rx + b"$"])
    n = rng.choice([17, 100, 2048, 2049, 8192, 8200, 16384, 40000, 70001, 300000])
    dense = rng.random()
    if dense < 0.3:
        t = bytearray(rng.choices(alphabet, k=n))
    else:  # long runs: mostly L bytes, a break every now and then
        lb = [c for c in alphabet if c in lset] or alphabet
        t = bytearray(rng.choices(lb, k=n))
        for _ in range(rng.choice([0, 1, 3, 20, n // 500 + 1])):
            t[rng.randrange(n)] = rng.choice(alphabet)
    text = bytes(t)
    full = oracle.match_all(rx, text)
    mode = rng.random()
    kw = {}
    asserted = rx.startswith(b"^") or rx.endswith(b"$")

    def line_start(p):
        # A range begins afresh (no selection state is carried in) but its CONTEXT is the real text's: the byte before it says whether
        # it begins at a line start.  The oracle knows "the whole text" and "the suffix as a text of its own"; the two agree with the
        # engine on `^` patterns when the range begins at a line start, so such ranges are moved to one.
        while asserted and 0 < p < n and text[p - 1] not in b"\n\r":
            p += 1
        return min(p, n - 1) if n else 0
    if mode < 0.5:
        want = full
    elif mode < 0.75:
        ob = line_start(rng.randrange(0, n)); oe = rng.randrange(ob, n + 2)
        kw = dict(own_begin=ob, own_end=oe)
        want = [(x + ob, y + ob) for x, y in oracle.match_all(rx, text[ob:]) if x + ob < oe]
    else:
        cut = line_start(rng.randrange(0, n))
        before = [m for m in full if m[0] < cut]
        kw = dict(own_begin=cut, own_end=n + 1)
        if before:
            bb, ee = before[-1]
            kw.update(carry_cur=ee if ee > bb else bb + 1, carry_prev_end=ee, have_prev=True)
            want = [m for m in full if m[0] >= cut]
        else:
            want = [(x + cut, y + cut) for x, y in oracle.match_all(rx, text[cut:])]
    try:
        prog = rejit_amd.Program(rx)
        sc = rejit_amd.Scan(prog)
        d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        k = sc.run(d.data_ptr(), n, **kw)
        got = sc.spans()
        st = sc.stats()
    except rejit_amd.RejitError as e:
        print("ERROR", rx, e, flush=True)
        bad += 1
        continue
    took += 1 if st["run_path"] else 0
    refused += 0 if st["run_path"] else 1
    # (`^` / `$`: the kernels read the byte before the range -- a range that begins inside a line has no line start there, the
    # whole text's answer; an independent run of the suffix would see one)
    if got != want and "own_begin" in kw and "have_prev" not in kw and (not st["run_path"] or rx.startswith(b"^") or rx.endswith(b"$")):
        # (an independent range through a kernel that looks at the byte before the range -- match_small, dense_streams: the
        # whole text's matches that begin in the range; tests/test_gpu_runs.py accepts both)
        want = [m for m in full if kw["own_begin"] <= m[0] < kw["own_end"]]
    if st["exact_path"] and "own_begin" in kw:
        # a pattern at risk of the reference's ring artefact (`X+` / `X+ B` with an assertion) over a RANGE: the range owns whole segments
        # between the reference's synchronisation points, not "the matches that begin in it" (engine.hip: run_exact; tests/test_gpu_parity.py)
        continue
    if got != want or k != len(want):
        bad += 1
        print("MISMATCH", rx, "n", n, kw, "run_path", st["run_path"], "got", len(got), got[:3], "want", len(want), want[:3], flush=True)
        if os.environ.get("FUZZ_DUMP"):   # the case for a replay: <dir>/case_<k>.{rx,txt,kw}
            d = os.environ["FUZZ_DUMP"]
            os.makedirs(d, exist_ok=True)
            open(os.path.join(d, "case_%d.rx" % case), "wb").write(rx)
            open(os.path.join(d, "case_%d.txt" % case), "wb").write(text)
            open(os.path.join(d, "case_%d.kw" % case), "w").write(repr(kw) + "\n" + repr(st) + "\n")
print("cases %d: mismatches %d; run kernels %d, other paths %d" % (cases, bad, took, refused))
