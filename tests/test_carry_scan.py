"""CPU tests of the linear-time carry scan (rejit_amd/csrc/carry_scan.h).

The header is compiled with g++ into a TEST-ONLY driver (tests/support/carry_exec.cc) that runs the
same per-sub-chunk bodies the HIP kernels call -- summaries with a symbolic entering state, the
resolve pass, the emit pass, the chain selection -- one sub-chunk after the other, and the result
is compared with the oracle on every golden vector for several sub-chunk sizes (1 byte per
sub-chunk makes every byte a boundary).  Where the reference's ring artefact (Q8)
applies, the documented semantics (Oracle.match_all_spec) are the expectation, as for the parallel
verifier.
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
CSRC = os.path.join(ROOT, "rejit_amd", "csrc")
SO = os.path.join(HERE, "support", "libcarry_exec.so")
SRCS = [os.path.join(HERE, "support", "carry_exec.cc"), os.path.join(CSRC, "parser.cc"), os.path.join(CSRC, "lowering.cc")]
DEPS = SRCS + [os.path.join(CSRC, h) for h in ("carry_scan.h", "device_program.h", "lowering.h", "table_layout.h", "behind_walk.h", "lds_walk.h")]
_u64p = ctypes.POINTER(ctypes.c_uint64)


@pytest.fixture(scope="module")
def ce():
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(s) for s in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", SO] + SRCS)
    lib = ctypes.CDLL(SO)
    lib.ce_match_range.restype = ctypes.c_long
    lib.ce_match_range.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                   ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, _u64p, ctypes.c_uint64]
    lib.ce_match_all_behind.restype = ctypes.c_long
    lib.ce_match_all_behind.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, _u64p, ctypes.c_uint64]
    lib.ce_lds_walk_check.restype = ctypes.c_long
    lib.ce_lds_walk_check.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, _u64p]
    return lib


def carry(lib, rx, tx, sub, sb=0, se=None, cur=0, prev_end=0, have_prev=False):
    se = len(tx) + 1 if se is None else se
    cap = len(tx) + 2
    buf = (ctypes.c_uint64 * (2 * cap))()
    n = lib.ce_match_range(rx, tx, len(tx), sub, sb, se, cur, prev_end, int(have_prev), buf, cap)
    if n < 0:
        return int(n)
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]


def test_all_vectors(ce):
    oracle = Oracle()
    n = q8 = 0
    for rx, tx, exp_all, _ in V.all_matchall_cases():
        want = exp_all
        for sub in (1, 3, 16, 4096):
            got = carry(ce, rx, tx, sub)
            if got == -9:       # wider than 256 positions (the 1000-byte literals): not this driver's business
                assert len(rx) > 256
                break
            if got != want:
                spec = oracle.match_all_spec(rx, tx)
                assert spec != exp_all and got == spec, (rx, tx, sub, got, exp_all)
                want = spec
                q8 += 1
        n += 1
    assert n > 2500
    assert q8 <= 30


HARD = [b"[acgt]+", b"[^>]+", b"(ab|ba)+", b"a.*b", b"x*", b"a+b*", b"(a|b)*abb", b"[ab]+c|[bc]+d", b"^.*$", b".*x", b"\\d+x",
        b"[a-z]+@[a-z]+", b"a{2,}", b"(a|ab)(c|bcd)*", b"x*y?z+", b"(aa|aaa)+", b".{0,2}.", b"[ab]{1,30}c", b"a.{3,}b"]


def test_long_runs_vs_oracle(ce):
    """Unbounded repetitions over long runs (what the per-start verifier cannot do in linear time)."""
    oracle = Oracle()
    rng = random.Random(5)
    for rx in HARD:
        for alphabet, n in ((b"ab", 700), (b"acgt", 900), (b"abcx\n", 800), (b"aabb>xyz09@.cd", 1200)):
            tx = bytes(rng.choice(alphabet) for _ in range(n))
            spec = oracle.match_all_spec(rx, tx)
            for sub in (1, 7, 64, 256):
                assert carry(ce, rx, tx, sub) == spec, (rx, alphabet, sub)


def test_ranges_compose(ce):
    """Own ranges with the selection state carried over a cut reproduce the whole-text result."""
    oracle = Oracle()
    rng = random.Random(11)
    for rx in HARD + [b"^", b"$", b"^$", b"x?"]:
        tx = bytes(rng.choice(b"abx\nacgt") for _ in range(500))
        want = oracle.match_all_spec(rx, tx)
        for cut in (1, 64, 250, 333, 499, 500):
            first = carry(ce, rx, tx, 32, 0, cut)
            if first:
                b, e = first[-1]
                state = (e if e > b else b + 1, e, True)
            else:
                state = (0, 0, False)
            second = carry(ce, rx, tx, 32, cut, len(tx) + 1, *state)
            assert first + second == want, (rx, cut)


def test_wide_automata(ce):
    """More than 32 / 64 / 128 positions (2, 4, 8 state words)."""
    oracle = Oracle()
    rng = random.Random(3)
    for rx in (b"[ab]{40}c*", b"(abcdefghijklmnopqrstuvwxyz0123456789)+x*", b"[ab]{70,90}", b"([ab]{3}c){30,}", b"a{150}b*"):
        tx = bytes(rng.choice(b"ab") for _ in range(600)) + b"c" + b"a" * 200 + b"bbb"
        want = oracle.match_all_spec(rx, tx)
        assert not isinstance(want, int)
        for sub in (5, 128):
            assert carry(ce, rx, tx, sub) == want, (rx, sub)


def test_automata_of_up_to_1024_positions(ce):
    """More than 256 positions (16 and 32 state words; round 4 -- until then the one RJ_TOO_LARGE reachable at match time): cyclic
    automata whose candidates live for the whole text."""
    oracle = Oracle()
    rng = random.Random(9)
    words = ["".join(rng.choice("abcd") for _ in range(rng.randint(6, 10))) for _ in range(40)]
    more = ["".join(rng.choice("abcd") for _ in range(rng.randint(7, 9))) for _ in range(110)]
    for ws, lo, hi in ((words, 256, 512), (more, 512, 1024)):
        rx = ("(" + "|".join(ws) + ")+").encode()
        tx = "".join(rng.choice(ws) for _ in range(60)).encode() + b"x" + "".join(rng.choice(ws) for _ in range(25)).encode() + b"ab"
        want = oracle.match_all_spec(rx, tx)
        assert not isinstance(want, int) and len(want) >= 2 and want[0][1] - want[0][0] > 300
        for sub in (7, 64, 4096):
            got = carry(ce, rx, tx, sub)
            assert got == want, (len(rx), sub)
        cut = len(tx) // 2
        assert carry(ce, rx, tx, 32, 0, cut) == [m for m in want if m[0] < cut]


def test_automata_beyond_1024_positions(ce):
    """64, 128 and 256 state words (round 5: the carry scan takes every automaton the lowering accepts, 8192 positions; until then
    a cyclic automaton of more than 1024 positions with a long-lived candidate was RJ_TOO_LARGE at match time)."""
    oracle = Oracle()
    rng = random.Random(17)
    for n_words, subs in ((250, (64, 4096)), (480, (4096,)), (900, (4096,))):
        ws = ["".join(rng.choice("abcd") for _ in range(rng.randint(7, 9))) for _ in range(n_words)]
        rx = ("(" + "|".join(ws) + ")+").encode()
        tx = "".join(rng.choice(ws) for _ in range(50)).encode() + b"x" + "".join(rng.choice(ws) for _ in range(20)).encode() + b"ab"
        want = oracle.match_all_spec(rx, tx)
        assert not isinstance(want, int) and len(want) >= 2 and want[0][1] - want[0][0] > 300
        for sub in subs:
            assert carry(ce, rx, tx, sub) == want, (n_words, sub)
        cut = len(tx) // 2
        assert carry(ce, rx, tx, 512, 0, cut) == [m for m in want if m[0] < cut]


def test_behind_walk_device_code_vs_oracle(ce):
    """rejit_amd/csrc/behind_walk.h -- the per-hit procedure verify_behind_in_regions runs (forward check
    from the cut, reverse automaton to the left-most start, forward longest) -- compiled for the CPU:
    oracle's answer or a flagged conflict / walk limit, never a wrong result."""
    oracle = Oracle()
    rng = random.Random(23)
    pats = [b".*regexp", b"[a-z]+abcdefgh", b"\\d+regexp", b"[0-9]+x", b"[A-Z][a-z]+ [A-Z][a-z]+", b"[ab]*abb", b"^.*foo", b"[a-z]+@[a-z]+",
            b".*ab.*cd", b"a+(bc|bd)e*", b"[ab]+(c|dd)+x", b"[ab]{30,}cd", b"([ab]{3}c){12,}xy"]
    used = flagged = 0
    for rx in pats:
        for alphabet in (b"abregxp0\n", b"ab", b"abcdx \nAB@", b"abcdefgh12x"):
            for n in (9, 80, 300):
                tx = bytes(rng.choices(alphabet, k=n))
                for plant in (b"", b"regexp", b"abcdefgh", b"abb", b"cd", b"xy"):
                    t2 = tx[:n // 2] + plant + tx[n // 2:]
                    buf = (ctypes.c_uint64 * (2 * (len(t2) + 2)))()
                    for walk in (1 << 20, 16):
                        k = ce.ce_match_all_behind(rx, t2, len(t2), walk, buf, len(t2) + 2)
                        if k == -101:
                            break
                        if k in (-100, -102):
                            flagged += 1
                            continue
                        assert k >= 0, (rx, k)
                        got = [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(k)]
                        assert got == oracle.match_all_spec(rx, t2), (rx, t2, walk)
                        used += 1
    assert used > 500, (used, flagged)


def test_lds_walkers_equal_the_walkers_they_replace(ce):
    """rejit_amd/csrc/lds_walk.h (padded tables, rows by position: what verify_lds.hip walks in LDS) against
    device_program.h / behind_walk.h: the longest match from every start, the forward check and the backward walk from
    every (text position, automaton position), the behind candidate of every text position -- found / begin / end /
    overrun flags all equal, with and without a short walk limit, automata of one to four 32-bit words, with assertions."""
    rng = random.Random(41)
    pats = [b"regexp", b"[cgt]gggtaaa|tttaccc[acg]", b"([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)", b"[a-z]+abcdefgh", b".*regexp",
            b"^ab+c$", b"(^|x)a*b$", b"\\d+regexp", b"[ab]*abb", b"a+(bc|bd)e*", b"(abcdefghijklmnopqrstuvwxyz0123456789)+x*", b"(ab|ba)+", b"x*",
            b"(abc|abd|b+c?)(de)?$", b"^$", b"[a-f]+[0-9]", b"(a|b|c|d)(e|f)(g|h)+", b"(some|[stuff])((other|regexps)? bla root blah | (abcdefgh)+)"]
    total = 0
    for rx in pats:
        for alphabet in (b"ab", b"abcdegxprt\n", b"abcdefgh xy01\n\r"):
            for n in (0, 1, 48):
                tx = bytes(rng.choices(alphabet, k=n))
                for plant in (b"", b"regexpregexpabcdefghthe", b"abb", b"abcdefghat"):
                    t2 = tx[:n // 2] + plant + tx[n // 2:]
                    for walk in (1 << 20, 7):
                        checked = ctypes.c_uint64(0)
                        bad = ce.ce_lds_walk_check(rx, t2, len(t2), walk, ctypes.byref(checked))
                        assert bad == 0, (rx, t2, walk, bad)
                        total += checked.value
    # (lowering these takes 0.03-0.7 s each -- the ring-artefact analysis -- so one text apiece: 3 and 4 words, rows)
    for rx in (b"([ab]{3}c){12,}xy", b"[ab]{30,40}cd", b"[ab]{70,90}c"):
        tx = bytes(rng.choices(b"ab", k=150)) + b"cdxy"
        checked = ctypes.c_uint64(0)
        assert ce.ce_lds_walk_check(rx, tx, len(tx), 1 << 20, ctypes.byref(checked)) == 0, rx
        total += checked.value
    assert total > 300_000, total
