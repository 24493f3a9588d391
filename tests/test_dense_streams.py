"""CPU test of the bit-stream dense kernel's algorithm (rejit_amd/csrc/dense_streams.h + table_layout.h: make_stream_plan):
the header is compiled with g++ into tests/support/carry_exec.cc, which runs it lane by lane exactly as
dense_streams.hip does -- 32 bytes per lane, the start frame 16 bytes back, class streams from byte-parallel range tests,
position-major steps, the scalar walk for starts that outlive the plan's depth -- and returns the MatchAll result.
Checked: against the oracle (whole texts and own ranges), start by start against the scalar automaton (stats[2]), and
that the plan refuses every pattern whose candidates can overlap."""
import ctypes
import os
import random
import subprocess

import pytest

from checkers import Oracle
from test_carry_scan import SO, SRCS, DEPS, CSRC

DEPS = DEPS + [os.path.join(CSRC, h) for h in ("dense_streams.h", "dense_swar.h", "exact_replay.h")]
_u64p = ctypes.POINTER(ctypes.c_uint64)


@pytest.fixture(scope="module")
def ce():
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(s) for s in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", SO] + SRCS)
    lib = ctypes.CDLL(SO)
    lib.ce_stream_match_all.restype = ctypes.c_long
    lib.ce_stream_match_all.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _u64p, ctypes.c_uint64, _u64p]
    return lib


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def stream_match_all(ce, rx, text, sb=0, se=None):
    n = len(text)
    if se is None:
        se = n + 1
    cap = n + 2
    out = (ctypes.c_uint64 * (2 * cap))()
    stats = (ctypes.c_uint64 * 8)()
    k = ce.ce_stream_match_all(rx, text, n, sb, se, out, cap, stats)
    if k < 0:
        return k, None, stats
    return k, [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(k)], stats


QUALIFY = [b"[a-f]+[0-9]", b"[@#]", b"[a-h][i-p]", b"[a-h]+[i-p]", b"[a-p]", b"[a-p]+", b"[0-9]+x", b"[^a-z]", b"[a-cx-z0-3]+q",
           b"[\x80-\xff]+a", b"[^ -~]+", b"[a-f]+[0-9]+[g-k]", b"abc[0-9]", b"a[b-d]e", b"(ab|cd)", b"[ab]c|[de]f|g", b"[a-f]+[0-9][^a-f0-9]",
           b"x[0-9]+y"]
# candidates can overlap, matches of at most 16 bytes: taken since round 5, the selection made in the kernel (StreamPlan::select)
SELECT = [b"[0-9][0-9][0-9]", b"..", b"a.c", b"[a-c]{4}d", b"(ab|ba)", b"[a-z][a-z][0-9]", b"[ab][bc][cd]", b"[0-9]{8}", b"[a-f]{8}", b"aa|aab", b"[ab]{2,3}"]
REFUSED = [b"[A-Z][a-z]+ [A-Z]", b"[a-f]+[0-9][a-f]", b"^[a-z]+", b"x*", b"(ab|cd)+e", b"ab?c", b"[a-z]+@[a-z]+",
           b"[a-f][0-9]?[a-f]?z", b".{0,2}.", b"[a-f]{17}"]


def test_stream_plan_takes_and_refuses(ce):
    text = b"abc def 123 xyz\n" * 4
    for rx in QUALIFY + SELECT:
        k, _, _ = stream_match_all(ce, rx, text)
        assert k >= 0, rx
    for rx in REFUSED:
        k, _, _ = stream_match_all(ce, rx, text)
        assert k == -101, rx


def test_stream_match_all_equals_oracle(ce, oracle):
    rng = random.Random(41)
    decided = walked = 0
    alphabets = [bytes(range(256)), b"abcdefxyz0123456789@# ABCq\n", b"ab0", bytes(range(0x60, 0x90)), b"az09{`/:AZ[@", b"abcdefghijklmnop", b"abcdefgh1"]
    for rx in QUALIFY:
        for alphabet in alphabets:
            for n in (0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 1000, 4099):
                tx = bytes(rng.choice(alphabet) for _ in range(n))
                k, got, st = stream_match_all(ce, rx, tx)
                assert st[2] == 0, (rx, alphabet[:8], n, "register steps differ from the scalar automaton")
                assert got == oracle.match_all(rx, tx), (rx, alphabet[:8], n)
                decided += st[0]
                walked += st[1]
    assert decided > 10000 and walked > 100, (decided, walked)


def test_stream_long_runs_and_ranges(ce, oracle):
    """Runs longer than the register depth (the scalar walk), matches across lane / chunk edges, own ranges of a sharded
    run (a match belongs to the range that holds its begin; the run-start rule looks at the byte before the range)."""
    rng = random.Random(43)
    for rx in (b"[a-f]+[0-9]", b"[a-p]+", b"[0-9]+x", b"[a-h]+[i-p]"):
        parts = []
        for _ in range(300):
            parts.append(bytes(rng.choice(b"abcdef") for _ in range(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 70, 200]))))
            parts.append(rng.choice([b"1", b"5 ", b"x", b" ", b"9z", b"pi", b"0123x"]))
        tx = b"".join(parts)
        want = oracle.match_all(rx, tx)
        k, got, st = stream_match_all(ce, rx, tx)
        assert st[2] == 0 and got == want, rx
        n = len(tx)
        for lo, hi in ((0, n // 3), (n // 3, n // 2), (n // 2, n + 1), (17, 18), (1000, 1032)):
            k, got, st = stream_match_all(ce, rx, tx, lo, hi)
            assert got == [m for m in want if lo <= m[0] < hi], (rx, lo, hi)


def test_refused_patterns_really_overlap(ce, oracle):
    """The plan's reason for refusing: for each refused chain pattern a text exists on which two candidates overlap (so the
    candidates are NOT the selection) -- the analysis is not merely timid on these."""
    cases = [(b"[A-Z][a-z]+ [A-Z]", b"Ab Cd Ef"), (b"[a-f]+[0-9][a-f]", b"ab1cd2e"), (b"[a-z]+@[a-z]+", b"ab@cd@ef")]
    for rx, tx in cases:
        all_starts = []
        for s in range(len(tx)):
            m = oracle.match_all(rx, tx[s:])
            if m and m[0][0] == 0:
                all_starts.append((s, s + m[0][1]))
        overlapping = any(a[1] > b[0] for a, b in zip(all_starts, all_starts[1:]))
        assert overlapping, rx


def test_selection_in_the_kernel_equals_oracle(ce, oracle):
    """Plans with `select` (candidates may overlap): the kernel's selection -- rj_stream_select inside a lane, speculation from
    lane to lane, a tile's entry state from the 2 KiB before it -- replayed on the CPU against the oracle.  Small texts (one
    tile), texts packed with matches (every lane holds one: many rounds of correction), own ranges."""
    rng = random.Random(47)
    alphabets = [b"ab0", b"abcd012", b"abcdefxyz0123456789@# ABCq\n", bytes(range(0x30, 0x7a)), b"0123456789", b"a"]
    most_rounds = 0
    for rx in SELECT:
        for alphabet in alphabets:
            for n in (0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 1000, 2047, 2048, 2049, 4099, 20000):
                tx = bytes(rng.choice(alphabet) for _ in range(n))
                k, got, st = stream_match_all(ce, rx, tx)
                assert k >= 0 and st[2] == 0, (rx, alphabet[:8], n, k)
                assert got == oracle.match_all(rx, tx), (rx, alphabet[:8], n)
                most_rounds = max(most_rounds, st[5])
                if n == 4099:
                    for lo, hi in ((0, 1500), (1500, 2050), (2050, n + 1), (17, 18), (2040, 2060)):
                        k, got, st = stream_match_all(ce, rx, tx, lo, hi)
                        # (an own range begins with nothing carried in: the selection restarts at `lo`)
                        assert k >= 0 and got == [(b + lo, e + lo) for b, e in oracle.match_all(rx, tx[lo:]) if b + lo < hi], (rx, lo, hi)
    assert most_rounds >= 20, most_rounds   # packed texts did need lane-after-lane correction


def test_selection_across_tiles(ce, oracle):
    """More than one 32-KiB tile.  Text whose matches are not packed: every tile finds a lane without a match in the 2 KiB
    before it and the result is the oracle's, matches across the tile edge included.  Text packed with matches: no such lane,
    the run is void (-103; the engine repeats it on scan_dense_walk) -- never a wrong result."""
    rng = random.Random(53)
    n = 3 * 32768 + 5000
    for rx in (b"[0-9][0-9][0-9]", b"[a-z][a-z][0-9]", b"(ab|ba)", b"[0-9]{8}"):
        sparse = bytearray(rng.choice(bytes(range(0x30, 0x7a))) for _ in range(n))
        for edge in (32768, 65536, 98304):            # packed stretches that END shortly before / run across a tile edge
            sparse[edge - 700:edge - 40] = bytes(rng.choice(b"0123456789ab") for _ in range(660))
            sparse[edge - 9:edge + 9] = b"12ab34ba5678901234"[:18]
        tx = bytes(sparse)
        k, got, st = stream_match_all(ce, rx, tx)
        assert k >= 0 and st[2] == 0, (rx, k)
        assert got == oracle.match_all(rx, tx), rx
        # own ranges that begin inside the second tile / end inside the third (a sharded run: the selection restarts at `lo`)
        for lo, hi in ((40000, 90000), (32768 - 5, 65536 + 5), (65536, n + 1)):
            k, got, st = stream_match_all(ce, rx, tx, lo, hi)
            assert k >= 0 and got == [(b + lo, e + lo) for b, e in oracle.match_all(rx, tx[lo:]) if b + lo < hi], (rx, lo, hi)
        packed = bytes(rng.choice(b"0123456789ab") for _ in range(n))
        k, got, st = stream_match_all(ce, rx, packed)
        assert k == -103 or got == oracle.match_all(rx, packed), (rx, k)
    # the whole text packed for the narrowest pattern: certainly void
    k, _, _ = stream_match_all(ce, b"..", bytes(rng.choice(b"xyz") for _ in range(n)))
    assert k == -103
