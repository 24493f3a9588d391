#!/usr/bin/env python3
"""CPU fuzz of the selection dense_streams makes for patterns whose candidates can overlap (StreamPlan::select, round 5): random
chain patterns (classes, literals, two-word alternations, {m,n} tails) through tests/support/carry_exec.cc -- which replays
the kernel's lane / iteration / tile logic -- against the oracle, on texts from packed with matches to sparse, one to three tiles.
usage: fuzz_select.py [cases] [seed]"""
import ctypes, os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from checkers import Oracle
from test_carry_scan import SO, SRCS

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
if not os.path.exists(SO):
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", SO] + SRCS)
lib = ctypes.CDLL(SO)
_u64p = ctypes.POINTER(ctypes.c_uint64)
lib.ce_stream_match_all.restype = ctypes.c_long
lib.ce_stream_match_all.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _u64p, ctypes.c_uint64, _u64p]
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


took = void = refused = bad = 0
for case in range(cases):
    rx = pattern()
    alphabet = rng.choice([LETTERS, LETTERS[:4], LETTERS + "xyzwvu   \n", LETTERS + "".join(chr(c) for c in range(0x40, 0x7f))])
    n = rng.choice([50, 700, 2100, 5000, 33000, 40000, 70000, 100000])
    text = "".join(rng.choice(alphabet) for _ in range(n)).encode()
    want = oracle.match_all(rx, text)
    if isinstance(want, int):
        continue
    cap = n + 2
    out = (ctypes.c_uint64 * (2 * cap))()
    stats = (ctypes.c_uint64 * 8)()
    k = lib.ce_stream_match_all(rx, text, n, 0, n + 1, out, cap, stats)
    if k == -101 or k < -103 or (k < 0 and k > -100):
        refused += 1
        continue
    if k == -103:
        void += 1
        continue
    took += 1
    got = [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(k)]
    if got != want or stats[2] != 0:
        bad += 1
        print("MISMATCH", rx, n, len(alphabet), len(got), len(want), stats[2], flush=True)
print("cases %d: answered %d (mismatches %d), void (packed text, more than one tile) %d, not a stream plan %d" % (cases, took, bad, void, refused))
