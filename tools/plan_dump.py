#!/usr/bin/env python3
"""CPU only: the scan plan (mode, windows, offsets, floating / behind, lengths) the host lowering chooses for a corpus of
patterns -- every regex of the golden fixtures, random ones in the fixture generator's style plus wider repetitions
and classes -- as JSON lines, and the time the lowering took.  Two dumps (before / after a change of lowering.cc)
must be identical when the change is meant to be a pure speed-up.
usage: plan_dump.py out.jsonl [random patterns, default 4000]"""
import ctypes, json, os, random, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vectors as V

CSRC = os.path.join(ROOT, "rejit_amd", "csrc")
so = "/tmp/libprogram_exec_plan.so"
subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + CSRC, "-o", so, os.path.join(ROOT, "tests", "support", "program_exec.cc"),
                       os.path.join(CSRC, "lowering.cc"), os.path.join(CSRC, "parser.cc")])
lib = ctypes.CDLL(so)
lib.pe_plan.restype = ctypes.c_int
lib.pe_plan.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]

corpus = []
seen = set()
def add(rx):
    if rx not in seen and b"\0" not in rx:
        seen.add(rx); corpus.append(rx)
for rx, _, _, _ in V.all_matchall_cases(): add(rx)
for rx, _, _, _ in V.artefact_cases(): add(rx)
for size in V.bench().values():
    for cases in (size.values() if isinstance(size, dict) else []):
        for p in (cases.get("patterns", []) if isinstance(cases, dict) else []):
            add(V.b(p["regex"]))
rng = random.Random(99)
n_random = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
def atom(depth):
    k = rng.random()
    if k < 0.4: return "".join(rng.choice("abcdxyz01 _") for _ in range(rng.randint(1, 9)))
    if k < 0.5: return "."
    if k < 0.7:
        body = "".join(rng.sample("abcdefxyz0123", rng.randint(1, 4)))
        if rng.random() < 0.4: body = rng.choice(["a-f", "0-9", "a-z", "x-z0-3"]) + body[:1]
        return "[" + ("^" if rng.random() < 0.15 else "") + body + "]"
    if k < 0.75: return rng.choice(["^", "$"])
    if depth <= 0: return rng.choice("abc")
    return "(" + "|".join(concat(depth - 1) for _ in range(rng.choice([1, 2, 2, 3]))) + ")"
def quant(depth):
    a = atom(depth)
    if rng.random() < 0.55: return a
    return a + rng.choice(["*", "+", "?", "{2}", "{1,3}", "{0,2}", "{2,}", "{3,9}", "{8}", "{2,7}", "{12,20}", "{30,40}", "{0,40}", "{64}"])
def concat(depth): return "".join(quant(depth) for _ in range(rng.randint(1, 4)))
for _ in range(n_random): add(concat(2).encode())
for rx in [b"[ab]{30,40}cd", b"[ab]{100}", b"([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)", b"(ab|cd){20,30}e", b".{0,60}needle", b"[acgt]{8}", b"[acgt]{12}x"]:
    add(rx)

t_all = time.perf_counter()
slow = []
with open(sys.argv[1], "w") as f:
    for rx in corpus:
        info = (ctypes.c_uint64 * 16)(); vals = (ctypes.c_uint32 * 64)()
        t0 = time.perf_counter()
        st = lib.pe_plan(rx, info, vals)
        dt = time.perf_counter() - t0
        if dt > 0.02: slow.append((round(dt * 1e3, 1), rx.decode("latin1")))
        rec = {"rx": rx.decode("latin1"), "st": st}
        if st == 0:
            rec["info"] = [int(x) for x in info[:12]]
            rec["win"] = [int(v) for v in vals[:4 * int(info[1])]]
        f.write(json.dumps(rec) + "\n")
print(len(corpus), "patterns,", round(time.perf_counter() - t_all, 2), "s; slower than 20 ms:", len(slow))
for s in sorted(slow, reverse=True)[:12]: print("  ", s)
