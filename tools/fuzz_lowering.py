#!/usr/bin/env python3
"""CPU only: the host lowering (parser, position automaton, scan plan) and the CPU mirror of the device pipeline
(tests/support/program_exec.cc: candidates from the plan's windows, verification by the automaton, left-most-longest
selection; the behind-window variant too) against the oracle, on fresh random patterns and texts -- the generator of the
golden fixtures plus wider repetitions -- beyond the 3819 committed vectors.  A difference is acceptable only where the
reference's ring artefact applies (the oracle's documented-semantics answer then has to be ours).
usage: fuzz_lowering.py [seed] [cases]"""
import ctypes, os, random, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from checkers import Oracle
from make_golden import RegexGen, ALPHABETS, ALPHABETS_HI

CSRC = os.path.join(ROOT, "rejit_amd", "csrc")
so = "/tmp/libprogram_exec_fuzz.so"
subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + CSRC, "-o", so, os.path.join(ROOT, "tests", "support", "program_exec.cc"),
                       os.path.join(CSRC, "lowering.cc"), os.path.join(CSRC, "parser.cc")])
lib = ctypes.CDLL(so)
_u64p = ctypes.POINTER(ctypes.c_uint64)
for f in (lib.pe_match_all, lib.pe_match_all_behind):
    f.restype = ctypes.c_long
    f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, _u64p, ctypes.c_uint64]
lib.pe_match_full.restype = ctypes.c_int
lib.pe_match_full.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64]

def run(fn, rx, tx):
    cap = len(tx) + 2
    buf = (ctypes.c_uint64 * (2 * cap))()
    n = fn(rx, tx, len(tx), buf, cap)
    return int(n) if n < 0 else [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = random.Random(seed)
o = Oracle()
ALPH = ALPHABETS + ["ab\n\r", "xyz ^$", "aA0-", "abcdefgh12 "]
if os.environ.get("FUZZ_HIGH_BYTES"):   # round 5: alphabets with bytes >= 0x80 (signed bracket ranges, Latin-1 / UTF-8 text)
    ALPH = [a.replace("\x00", "") for a in ALPHABETS_HI] + ["ab\x80\xff\n", "\xe9\xe8 ^$"]
checked = bad = q8 = behind = too_large = 0
t0 = time.time()
for it in range(cases):
    alphabet = rng.choice(ALPH)
    rx = RegexGen(rng, alphabet).alt(3)
    if rng.random() < 0.3:   # wider repetitions and a literal tail: window search, floating and behind plans
        rx += rng.choice(["{3,9}", "{8}", "{12,20}", "+", ""]) + "".join(rng.choice(alphabet.replace("\n", "").replace("\r", "") or "a") for _ in range(rng.randint(0, 9)))
    rx = rx.encode("latin1")
    if b"\0" in rx or o.status(rx) != 0:
        continue
    tx = "".join(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 9, 33, 120, 700, 2500]))).encode("latin1")
    want = o.match_all(rx, tx)
    if isinstance(want, int):
        continue
    got = run(lib.pe_match_all, rx, tx)
    if got == -2:          # RJ_TOO_LARGE: beyond the automaton limits (a documented limit, not an answer)
        too_large += 1
        continue
    checked += 1
    ok = got == want
    if not ok:
        spec = o.match_all_spec(rx, tx)
        if spec != want and got == spec:
            q8 += 1
            ok = True
    if ok:
        gb = run(lib.pe_match_all_behind, rx, tx)
        if isinstance(gb, list):        # (a negative status: the plan has no behind window)
            behind += 1
            ok = gb == got
    wf = o.match_full(rx, tx) if hasattr(o, "match_full") else None
    if ok and wf is not None and lib.pe_match_full(rx, tx, len(tx)) != wf:
        ok = False
    if not ok:
        bad += 1
        if bad <= 10:
            print("MISMATCH", rx, tx[:80], "got", (got if isinstance(got, int) else got[:4]), "want", want[:4], flush=True)
print(f"seed {seed}: checked {checked}, too large {too_large}, ring-artefact cases {q8}, behind plans {behind}, mismatches {bad}, {time.time() - t0:.0f}s")
sys.exit(1 if bad else 0)
