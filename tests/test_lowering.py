"""CPU tests of the host lowering (parser -> NFA graph -> position automaton -> scan plan).

The lowered Program is executed by a TEST-ONLY scalar executor (tests/support/
program_exec.cc, compiled here with g++) that applies exactly the tables the HIP kernels
interpret, and the result is compared with the oracle on every golden vector.

Known, documented divergence ("Q8"): on a handful of random vectors the
reference's own no-fast-forward loop drops a thread that starts exactly at the end of a
match; there the product implements the documented left-most-longest semantics
(Oracle.match_all_spec) and the reference's default-flag build returns a third,
overlapping answer.
"""
import ctypes
import os
import random
import subprocess

import pytest

import vectors as V
from checkers import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "support", "libprogram_exec.so")
SRCS = [os.path.join(HERE, "support", "program_exec.cc"),
        os.path.join(ROOT, "rejit_amd", "csrc", "parser.cc"),
        os.path.join(ROOT, "rejit_amd", "csrc", "lowering.cc")]
_u64p = ctypes.POINTER(ctypes.c_uint64)


@pytest.fixture(scope="module")
def pe():
    deps = SRCS + [os.path.join(ROOT, "rejit_amd", "csrc", "lowering.h")]
    deps.append(os.path.join(ROOT, "rejit_amd", "csrc", "lowering.cc"))
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(s) for s in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", SO] + SRCS)
    lib = ctypes.CDLL(SO)
    lib.pe_match_all.restype = ctypes.c_long
    lib.pe_match_all.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, _u64p, ctypes.c_uint64]
    lib.pe_match_full.restype = ctypes.c_int
    lib.pe_match_full.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64]
    lib.pe_match_all_behind.restype = ctypes.c_long
    lib.pe_match_all_behind.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, _u64p, ctypes.c_uint64]
    lib.pe_plan.restype = ctypes.c_int
    lib.pe_plan.argtypes = [ctypes.c_char_p, _u64p, ctypes.POINTER(ctypes.c_uint32)]
    return lib


def match_all(lib, rx, tx):
    cap = len(tx) + 2
    buf = (ctypes.c_uint64 * (2 * cap))()
    n = lib.pe_match_all(rx, tx, len(tx), buf, cap)
    if n < 0:
        return int(n)
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]


def plan(lib, rx):
    info = (ctypes.c_uint64 * 16)()
    vals = (ctypes.c_uint32 * 32)()
    st = lib.pe_plan(rx, info, vals)
    assert st == 0
    return dict(windows=bool(info[0]), n_windows=int(info[1]), offset=int(info[2]), n_pos=int(info[3]),
                min_len=int(info[4]), max_len=int(info[5]), n_rows=int(info[6]), literal_len=int(info[7]),
                floating=bool(info[8]), float_min=int(info[9]), float_max=int(info[10]), behind=bool(info[11]),
                values=[tuple(int(v) for v in vals[4 * i:4 * i + 4]) for i in range(int(info[1]))])


def test_all_vectors_match_reference(pe):
    oracle = Oracle()
    n = q8 = 0
    for rx, tx, exp_all, exp_full in V.all_matchall_cases():
        got = match_all(pe, rx, tx)
        if got != exp_all:
            # only acceptable on the Q8 artefact, where documented semantics != reference ff=0
            spec = oracle.match_all_spec(rx, tx)
            assert spec != exp_all and got == spec, (rx, tx, got, exp_all)
            q8 += 1
        assert pe.pe_match_full(rx, tx, len(tx)) == exp_full, (rx, tx)
        n += 1
    assert n > 2500
    assert q8 <= 25, q8   # 19 of the 2500 fuzz vectors (+ a few of the short high-byte ones); none of the test.cc-derived vectors


def test_high_byte_vectors_match_reference(pe):
    """The lowering (parser's signed bracket ranges, class tables, window plans over bytes >= 0x80) + the CPU mirror of
    the device pipeline against the real reference's outputs on the high-byte vectors (texts up to 5000 bytes)."""
    oracle = Oracle()
    n = q8 = 0
    for rx, tx, exp_all, exp_full in V.highbyte_cases():
        got = match_all(pe, rx, tx)
        if got != exp_all:
            spec = oracle.match_all_spec(rx, tx)      # (the ring artefact: documented semantics != reference)
            assert spec != exp_all and got == spec, (rx, tx[:80], got if isinstance(got, int) else got[:5])
            q8 += 1
        assert pe.pe_match_full(rx, tx, len(tx)) == exp_full, (rx, tx[:80])
        n += 1
    assert n >= 1800
    assert q8 <= 40, q8


def test_parse_errors(pe):
    for e in V.semantics()["errors"]:
        buf = (ctypes.c_uint64 * 4)()
        assert pe.pe_match_all(V.b(e["regex"]), b"abc", 3, buf, 2) == -1, e


def _w(s: bytes, wild=()):
    """(value0, mask0, value1, mask1) of an up-to-8-byte window; `wild` = wildcard positions."""
    v = [0, 0]
    m = [0, 0]
    for k, c in enumerate(s):
        if k in wild:
            continue
        v[k // 4] |= c << (8 * (k % 4))
        m[k // 4] |= 0xFF << (8 * (k % 4))
    return (v[0], m[0], v[1], m[1])


def test_scan_plans(pe):
    p = plan(pe, b"regexp")
    assert p["windows"] and p["values"] == [_w(b"regexp")] and p["offset"] == 0
    assert p["literal_len"] == 6 and p["min_len"] == 6 and p["max_len"] == 6
    dna = [V.b(x["regex"]) for x in V.bench()["regexdna"]["1000"]["patterns"]]
    p = plan(pe, dna[0])
    assert sorted(p["values"]) == sorted([_w(b"agggtaaa"), _w(b"tttaccct")])
    p = plan(pe, dna[1])   # [cgt]gggtaaa|tttaccc[acg]: the class bytes become wildcards
    assert sorted(p["values"]) == sorted([_w(b"?gggtaaa", (0,)), _w(b"tttaccc?", (7,))])
    for rx in dna:
        p = plan(pe, rx)
        assert p["windows"] and p["n_windows"] == 2, (rx, p)
        assert p["min_len"] == 8 and p["max_len"] == 8 and p["n_pos"] == 16
    p = plan(pe, b"x*")
    assert not p["windows"] and p["min_len"] == 0 and p["max_len"] == 2 ** 64 - 1
    p = plan(pe, b"([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)")
    assert p["min_len"] == 12 and p["max_len"] == 58
    # the reference's FF element `abcdefgh` floats 2..42 bytes after the start of the match
    assert p["windows"] and p["floating"] and p["values"] == [_w(b"abcdefgh")]
    assert (p["float_min"], p["float_max"]) == (2, 42)
    p = plan(pe, b"[a-z]+@example")
    # unbounded prefix: the literal behind it is the window, the start is found by the backward pass
    assert p["windows"] and p["behind"] and not p["floating"] and p["values"] == [_w(b"@example")]
    p = plan(pe, b"[0-9]{2,3}foo(bar|baz)")
    assert p["windows"] and p["floating"] and (p["float_min"], p["float_max"]) == (2, 3) and p["values"] == [_w(b"foo")]
    p = plan(pe, b">.*\n|\n")
    assert p["windows"] and p["n_windows"] == 2 and p["min_len"] == 1
    p = plan(pe, b"abc.efgh")      # a wide class inside the window is a wildcard, not a split
    assert p["windows"] and p["values"] == [_w(b"abc?efgh", (3,))]
    p = plan(pe, b"(alternation|more|than|two|different|strings)")
    assert p["windows"] and p["n_windows"] == 6 and p["min_len"] == 3


def match_all_behind(lib, rx, tx):
    cap = len(tx) + 2
    buf = (ctypes.c_uint64 * (2 * cap))()
    n = lib.pe_match_all_behind(rx, tx, len(tx), buf, cap)
    if n < 0:
        return int(n)
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]


def test_windows_behind_an_unbounded_prefix(pe):
    """`.*regexp`, `[a-z]+abcdefgh`, `\\d+regexp`: the reference fast-forwards on ANY literal and runs its
    NFA backwards from the hit (src/codegen.cc:352-383, codegen-x64.cc:643-650).  The plan picks such
    literals as windows with the automaton positions they enter at; the per-hit procedure (forward
    check from the cut, reverse automaton to the left-most start, forward longest) is restated by the
    test executor and must give the oracle's answer -- or report a conflict, never a wrong result."""
    import random
    oracle = Oracle()
    pats = [b".*regexp", b"[a-z]+abcdefgh", b"\\d+regexp", b"[0-9]+x", b"[A-Z][a-z]+ [A-Z][a-z]+", b"(ab|ba)+c", b"a.*b",
            b"[ab]*abb", b"(x|yy)+z[ab]*", b"^.*foo", b"[a-z]+@[a-z]+", b".*ab.*cd", b"(a|b)+c(a|b)+", b"a+(bc|bd)e*", b"[ab]+(c|dd)+x"]
    rng = random.Random(17)
    used = conflicts = 0
    for rx in pats:
        pl = plan(pe, rx)
        if not pl["behind"]:     # a fixed-offset window set was found instead: not this test's business
            assert rx not in (b".*regexp", b"[a-z]+abcdefgh", b"\\d+regexp", b"[0-9]+x", b"[a-z]+@[a-z]+"), (rx, pl)
            continue
        assert pl["windows"]
        for alphabet in (b"abregxp0\n", b"ab", b"abcdx \nAB@", b"abcdefgh12x", b"abcde"):
            for n in (7, 60, 400):
                tx = bytes(rng.choices(alphabet, k=n))
                for plant in (b"", b"regexp", b"abcdefgh", b"abb", b"zz"):
                    t2 = tx[:n // 2] + plant + tx[n // 2:]
                    got = match_all_behind(pe, rx, t2)
                    want = oracle.match_all_spec(rx, t2)
                    if got == -100:
                        conflicts += 1
                        continue
                    assert got == want, (rx, t2, got, want)
                    used += 1
    assert used > 500 and conflicts < used // 10, (used, conflicts)


def test_run_start_rule_is_sound(pe):
    """`X+ rest` patterns: the dense kernel takes only the first byte of a run of X as a candidate
    (DevProgram::loop_first, table_layout.h: run_start_rule).  That is only right when no longest match can
    end inside a run -- `[a-f]+[0-9][a-f]` over "ab1cd2e" has the matches (0,4) and (4,7), the second one
    beginning in the middle of the run "cd".  The executor must give the documented semantics on all of
    these (the reference's own answer differs on some by its ring artefact, which the engine's exact replay
    handles separately)."""
    oracle = Oracle()
    rng = random.Random(21)
    pats = [b"[a-f]+[0-9][a-f]", b"[a-z]+@[a-z]+", b"[a-f]+[0-9]", b"a+ba", b"[ab]+a", b"[ab]+b[ab]", b"x+yx?", b"[0-9]+x", b"[a-c]+[b-d]",
            b"[a-c]+d[a-c]*", b"a+(b|ca)", b"[ab]+c?[ab]"]
    assert match_all(pe, b"[a-f]+[0-9][a-f]", b"ab1cd2e") == [(0, 4), (4, 7)]
    for rx in pats:
        for alphabet in (b"ab1cd2e", b"ab", b"abcd", b"xy", b"abc@.", b"a1b2"):
            for _ in range(40):
                tx = bytes(rng.choice(alphabet) for _ in range(rng.choice([7, 30, 120])))
                assert match_all(pe, rx, tx) == oracle.match_all_spec(rx, tx), (rx, tx)


def test_compile_time_of_wide_repetitions(pe):
    """The window search merges candidate strings pairwise; done naively that was cubic and `[ab]{30,40}cd` took half a
    second to compile ([ab]{8} = 256 strings at 40 window offsets).  Generous bounds: the point is the order of growth."""
    import time
    for rx, limit in ((b"[ab]{30,40}cd", 0.25), (b"[ab]{100}", 0.4), (b"([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)", 0.1),
                      (b"[acgt]{12}x", 0.25)):
        t0 = time.perf_counter()
        p = plan(pe, rx)
        dt = time.perf_counter() - t0
        assert dt < limit, (rx, dt)
        assert p["n_pos"] > 0
