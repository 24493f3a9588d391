#!/usr/bin/env python3
"""Parity fuzz of the LDS verify tails (verify_lds.hip) on the GPU box: random patterns whose plan is a window behind an
unbounded prefix or a floating window -- with and without `^` / `$`, one to three 64-bit state words -- over texts that
are dense in hits and near-hits (dozens of hits per region, walks of very different lengths, matches that end at line
breaks and at the end of the text), through the general pipeline (RJ_NO_SMALL=1 set below).
usage: fuzz_verify_lds.py [cases] [seed]"""
import os, random, sys
os.environ.setdefault("RJ_NO_SMALL", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rejit_amd
from checkers import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
o = Oracle()
bad = checked = 0
modes = {}
LITS = ["xyz", "regexp", "abcdefgh", "qq", "needle", "@", "ab1"]
for trial in range(cases):
    lit = rng.choice(LITS)
    kind = rng.choice(["behind", "behind", "floating"])
    if kind == "behind":
        pre = rng.choice(["[a-z]+", ".*", "[0-9]+", "[ab]*", "(ab|c)+", "^.*", "[a-c]+x?", "[a-z]+[0-9]*", "(a|bc)*d?"])
        suf = rng.choice(["", "[a-z]+", "(x|yz)", "$", "[0-9]*", ".*$", "[ab]{0,3}", "(a+|b)c?"])
    else:
        pre = rng.choice(["[ab]{0,3}", "[ab]{1,4}", "(a|bb){1,3}", "a?b?", "(ab|b){2,5}", "[ab]{2,7}c?", "([ab]|cc){1,6}", "^[ab]{0,5}", "(abc|de){1,9}"])
        suf = rng.choice(["", "[ab]", "(a|bc)", "b{0,2}", "[ab]+", "$", "[ab]*$", "(a|b|c){0,40}"])
    rx = (pre + lit + suf).encode()
    n = rng.choice([300, 700, 5000, 40000])
    alphabet = rng.choice(["abcxyz \n", "ab", "abcdefgh12x\n", "abq@1 \r\n", "regxpab01\n"])
    parts = []
    while sum(map(len, parts)) < n:
        r = rng.random()
        if r < 0.2:
            parts.append("".join(rng.choice("abc") for _ in range(rng.randrange(0, 60))) + lit + rng.choice(["", "a", "bc", "bb", "\n", "xyz"]))
        elif r < 0.3:
            parts.append(lit[:-1])
        else:
            parts.append("".join(rng.choice(alphabet) for _ in range(rng.randrange(1, 40))))
    text = ("".join(parts)[:n] + rng.choice(["", lit, "ab" + lit, lit + "ab"])).encode()
    want = o.match_all(rx, text)
    if isinstance(want, int):
        continue
    spec = o.match_all_spec(rx, text)
    try:
        p = rejit_amd.Program(rx)
    except rejit_amd.RejitError:
        continue
    info = p.info()
    got = p.match_all(text)
    checked += 1
    key = (kind, info["scan_mode"], info["n_words"], info["has_assertions"])
    modes[key] = modes.get(key, 0) + 1
    if got != want and got != spec:
        bad += 1
        if bad <= 8:
            sg, sw = set(got), set(spec)
            print("MISMATCH", rx, "n", len(text), info, "extra", sorted(sg - sw)[:3], "missing", sorted(sw - sg)[:3], flush=True)
print("checked", checked, "bad", bad, "plans", sorted(modes.items()))
sys.exit(1 if bad else 0)
