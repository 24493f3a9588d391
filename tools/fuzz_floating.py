#!/usr/bin/env python3
"""One-off: patterns with a required literal behind a bounded variable-length prefix (floating
windows) over texts dense in near-matches, vs the oracle."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rejit_amd
from checkers import Oracle
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
o = Oracle()
bad = floating = checked = 0
for trial in range(400):
    lit = "".join(rng.choice("xyz") for _ in range(rng.choice([3, 5, 8, 10])))
    pre = rng.choice(["[ab]{0,3}", "[ab]{1,4}", "(a|bb){1,3}", "a?b?", "(ab|b){2,5}", "[ab]{2,7}c?", "([ab]|cc){1,6}"])
    suf = rng.choice(["", "[ab]", "(a|bc)", "b{0,2}", "[ab]+"])
    rx = (pre + lit + suf).encode()
    n = rng.choice([200, 3000, 20000])
    parts = []
    while sum(map(len, parts)) < n:
        r = rng.random()
        if r < 0.15: parts.append("".join(rng.choice("abc") for _ in range(rng.randrange(0, 9))) + lit + rng.choice(["", "a", "bc", "bb"]))
        elif r < 0.25: parts.append(lit[:-1])
        else: parts.append("".join(rng.choice("abcxyz ") for _ in range(rng.randrange(1, 12))))
    text = "".join(parts)[:n].encode()
    want = o.match_all(rx, text)
    if isinstance(want, int): continue
    p = rejit_amd.Program(rx)
    got = p.match_all(text)
    checked += 1
    info = p.info()
    floating += info["scan_mode"] == 1 and info["window_offset"] == 0 and info["min_len"] != info["window_len"]
    if got != want:
        bad += 1
        k = next((j for j in range(min(len(got), len(want))) if got[j] != want[j]), min(len(got), len(want)))
        print("MISMATCH", rx, n, k, got[k:k+2], want[k:k+2], "spec" if got == o.match_all_spec(rx, text) else "")
        if bad > 8: break
print("checked", checked, "mismatches", bad)
