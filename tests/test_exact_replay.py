"""CPU tests of the exact replay (rejit_amd/csrc/exact_replay.h): synchronisation points from the
all-starts position automaton, then the reference's own loop per segment.

Unlike the parallel verifier and the carry scan, this path must give the REFERENCE's answer on every
input, the ring artefact (Q8) included: the expectation is always the golden vector /
Oracle.match_all, never match_all_spec.  The header is compiled with g++ into the test-only driver
tests/support/carry_exec.cc, which drives it the way exact_replay.hip does (ownership by synchronisation
points, one proven point per chunk, one replay per segment).
"""
import ctypes
import os
import random
import subprocess

import pytest

import vectors as V
from checkers import Oracle
from test_carry_scan import SO, SRCS, DEPS, CSRC

_u64p = ctypes.POINTER(ctypes.c_uint64)
DEPS = DEPS + [os.path.join(CSRC, "exact_replay.h")]


@pytest.fixture(scope="module")
def ce():
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(s) for s in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", SO] + SRCS)
    lib = ctypes.CDLL(SO)
    lib.ce_exact_range.restype = ctypes.c_long
    lib.ce_exact_range.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                   ctypes.c_uint64, _u64p, ctypes.c_uint64, _u64p, _u64p, ctypes.POINTER(ctypes.c_int)]
    return lib


def exact(lib, rx, tx, chunk, sb=0, se=None):
    se = len(tx) + 1 if se is None else se
    cap = len(tx) + 2
    buf = (ctypes.c_uint64 * (2 * cap))()
    nseg, longest, risk = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
    n = lib.ce_exact_range(rx, tx, len(tx), chunk, sb, se, buf, cap, ctypes.byref(nseg), ctypes.byref(longest), ctypes.byref(risk))
    if n < 0:
        return int(n), None
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)], (nseg.value, longest.value, risk.value)


def test_all_vectors_equal_the_reference(ce):
    n = risky = 0
    for rx, tx, exp_all, _ in V.all_matchall_cases():
        for chunk in (1, 3, 16, 4096):
            got, info = exact(ce, rx, tx, chunk)
            if got == -9:
                assert len(rx) > 256
                break
            assert got == exp_all, (rx, tx, chunk, got, exp_all)
            risky += info[2]
        n += 1
    assert n > 2500 and risky > 0


def test_ring_artefact_vectors_equal_the_reference(ce):
    """The 840 artefact vectors (expectations from the real reference): the replay gives the reference's answer
    whole-text and split into ranges."""
    n = 0
    for rx, tx, exp_all, _ in V.artefact_cases():
        for chunk in (7, 1024):
            got, _info = exact(ce, rx, tx, chunk)
            assert got == exp_all, (rx, tx, chunk)
        cut = len(tx) // 2
        a, _ = exact(ce, rx, tx, 16, 0, cut)
        b, _ = exact(ce, rx, tx, 16, cut, len(tx) + 1)
        assert a + b == exp_all, (rx, tx, cut)
        n += 1
    assert n >= 800


AT_RISK = [b".{0,2}.", b"x*", b"(a|ab)(c|bcd)*", b"a*b*", b"(ab|a)*", b".{0,3}x?", b"[ab]*c?", b"(a|b|ab)+", b"^.{0,2}.", b"a?b?c?",
           b"(x|xy)*z?", b".?.?"]


def test_segments_are_independent_and_exact(ce):
    """Random texts with many adjacent candidates: whole text and every split into two / three ranges give the
    reference's answer; the segments really are many (the replay is parallel work, not one long run)."""
    oracle = Oracle()
    rng = random.Random(11)
    differs = total_segments = 0
    for rx in AT_RISK:
        for trial in range(6):
            alphabet = [b"abcxyz\n", b"ab", b"abc\n\r", b"xyz \n"][trial % 4]
            n = rng.choice([0, 1, 7, 64, 300, 1500])
            tx = bytes(rng.choice(alphabet) for _ in range(n))
            want = oracle.match_all(rx, tx)
            differs += want != oracle.match_all_spec(rx, tx)
            for chunk in (1, 5, 64, 1024):
                got, info = exact(ce, rx, tx, chunk)
                assert got == want, (rx, tx, chunk)
                total_segments += info[0]
            for cuts in ([n // 2], [n // 3, 2 * n // 3], [1], [n]):
                bounds = [0] + sorted(cuts) + [n + 1]
                parts = []
                for lo, hi in zip(bounds, bounds[1:]):
                    got, _ = exact(ce, rx, tx, 16, lo, hi)
                    parts += got
                assert parts == want, (rx, tx, cuts, parts, want)
    assert differs >= 3           # the artefact really is exercised
    assert total_segments > 1000


def test_long_stretch_without_synchronisation_point(ce):
    oracle = Oracle()
    tx = b"x" * 5000 + b"\n" + b"ab" * 700 + b"q"
    for rx in (b"x*", b".{0,2}.", b"(ab|a)*"):
        want = oracle.match_all(rx, tx)
        for chunk in (64, 1024):
            got, info = exact(ce, rx, tx, chunk)
            assert got == want, rx
    got, info = exact(ce, b"x*", tx, 64)
    assert info[1] >= 5000        # one segment holds the whole run of x


def _risk(lib, rx):
    buf = (ctypes.c_uint64 * 8)()
    a, b, r = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int(-1)
    rc = lib.ce_exact_range(rx, b"", 0, 16, 0, 1, buf, 4, ctypes.byref(a), ctypes.byref(b), ctypes.byref(r))
    return None if (rc < 0 and r.value == -1) else bool(r.value)


def test_refined_risk_flag_is_sound(ce):
    """Program::q8_risk = the closure criterion AND NOT "every younger thread is absorbed by the older one"
    (lowering.cc: threads_always_nested).  A pattern that is not flagged never goes to the exact replay, so
    on it the reference must agree with the documented semantics on every text."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import RegexGen, ALPHABETS
    oracle = Oracle()
    for rx, want in ((b"[a-f]+[0-9]", False), (b"[a-z]+", False), (b"[acgt]+", False), (b"[^>]+", False), (b"x*", False), (b"[0-9]+x", False),
                     (b".*regexp", False), (b"[a-f]+[0-9][a-f]", True), (b".{0,2}.", True), (b"(ab|ba)+", True), (b"agggtaaa|tttaccct", False)):
        assert _risk(ce, rx) == want, rx
    assert oracle.match_all(b"[a-f]+[0-9][a-f]", b"ab1cd2e") == [(0, 4)]          # the artefact ...
    assert oracle.match_all_spec(b"[a-f]+[0-9][a-f]", b"ab1cd2e") == [(0, 4), (4, 7)]  # ... on an `X+ rest` pattern
    rng = random.Random(77)
    checked = flagged = artefact = 0
    for _ in range(3000):
        alphabet = rng.choice(ALPHABETS)
        rx = RegexGen(rng, alphabet).alt(2).encode("latin1")
        if oracle.status(rx) != 0:
            continue
        risk = _risk(ce, rx)
        if risk is None:
            continue
        checked += 1
        flagged += risk
        for k in range(4):
            al = alphabet if k % 2 == 0 else rng.choice(ALPHABETS)
            tx = "".join(rng.choice(al) for _ in range(rng.choice([8, 40, 150]))).encode("latin1")
            want = oracle.match_all(rx, tx)
            if isinstance(want, int):
                break
            if want != oracle.match_all_spec(rx, tx):
                artefact += 1
                assert risk, (rx, tx)
    assert checked > 2500 and 0 < flagged < checked * 0.6 and artefact > 20


def test_long_segments_speculate_and_verify(ce):
    """exact_replay.h, long segments: parts of a segment replayed from a few bytes earlier with a free ring, the true ring
    walked over them, a ring no part was replayed from made a candidate and a further round, the sink applied afterwards --
    the same answer as the one sequential replay (= the reference's, golden vectors), for parts of 1 .. 64 bytes and
    warm-ups of 0 .. 64 bytes; with no warm-up at all further rounds are the rule (and tiny parts may be given up: the
    driver then replays sequentially), with one they are the exception."""
    ce.ce_exact_range_spec.restype = ctypes.c_long
    ce.ce_exact_range_spec.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                       ctypes.c_uint64, ctypes.c_uint64, _u64p, ctypes.c_uint64, _u64p]

    def spec(rx, tx, sub, warm, sb=0, se=None):
        se = len(tx) + 1 if se is None else se
        cap = len(tx) + 2
        buf = (ctypes.c_uint64 * (2 * cap))()
        fixed = ctypes.c_uint64()
        n = ce.ce_exact_range_spec(rx, tx, len(tx), 4096, sb, se, sub, warm, buf, cap, ctypes.byref(fixed))
        if n < 0:
            return int(n), 0
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)], fixed.value

    cases = rounds_cold = rounds_warm = given_up_warm = 0
    for rx, tx, exp_all, _ in V.all_matchall_cases():
        if len(rx) > 256:
            continue
        for sub, warm in ((1, 0), (2, 1), (3, 0), (5, 4), (7, 64), (64, 8)):
            got, fixed = spec(rx, tx, sub, warm)       # fixed = rounds beyond the first + 10^6 per segment given up
            if got == -9:
                break
            assert got == exp_all, (rx, tx, sub, warm, got, exp_all)
            if warm == 0:
                rounds_cold += fixed % 1000000
            elif warm >= 8:
                rounds_warm += fixed % 1000000
                given_up_warm += fixed // 1000000
        cases += 1
    assert cases > 2500 and rounds_cold > 1000, (cases, rounds_cold)
    # (with a warm-up the parts nearly always agree: further rounds are the exception, and no segment is given up)
    assert rounds_warm * 20 < rounds_cold and given_up_warm == 0, (rounds_warm, rounds_cold, given_up_warm)
    # longer texts, at-risk patterns, own ranges
    rng = random.Random(17)
    oracle = Oracle()
    for rx in (b".{0,2}.", b"(a|ab)(c|bcd)(d*)", b"[ab]{1,3}b", b"a.{0,3}$", b"(x|xy|xyz)+z?"):
        for _ in range(6):
            tx = bytes(rng.choice(b"abcdxyz\n") for _ in range(rng.choice([50, 700, 3000])))
            want = oracle.match_all(rx, tx)
            for sub, warm in ((16, 0), (64, 16), (257, 32)):
                got, _ = spec(rx, tx, sub, warm)
                assert got == want, (rx, tx[:40], sub, warm)
            lo = len(tx) // 3
            got, _ = spec(rx, tx, 32, 8, lo, len(tx) + 1)
            whole, _ = exact(ce, rx, tx, 4096, lo, len(tx) + 1)
            assert got == whole, (rx, lo)
    # a thread that lives for thousands of bytes (the match begins long before the part that reports it): the parts only need
    # the ring's ORDER PATTERN -- no further rounds with a warm-up, a few without, never given up
    for rx, tx in ((b"[xy]+z[xy]", b"x" * 5000 + b"zx xzy" + b"y" * 3000 + b"zy"), (b"[a-f]+[0-9][a-f]", b"abc" * 2000 + b"1cd2e"),
                   (b"(ab|ba)+", b"ab" * 4000 + b"c" + b"ba" * 10)):
        want = oracle.match_all(rx, tx)
        for sub, warm in ((16, 0), (64, 8), (257, 32)):
            got, fixed = spec(rx, tx, sub, warm)
            assert got == want and fixed < 6, (rx, sub, warm, fixed)
            assert warm == 0 or fixed <= 1, (rx, sub, warm, fixed)
    # patterns that match the empty string: the sink's filter looks at the entry before, which may belong to another part --
    # the join decides it for a part's first entry (xr_join); tiny parts put nearly every decision there
    for rx in (b".{0,2}", b"(a|ab)?(c|bcd)?", b"x*y?", b"[ab]{0,2}", b"(ab|b)*", b"a?$", b"^b?", b".{0,3}d?"):
        for _ in range(8):
            tx = bytes(rng.choice(b"abcdxy\n") for _ in range(rng.choice([20, 300, 1500])))
            want, _ = exact(ce, rx, tx, 4096)
            assert want == oracle.match_all(rx, tx), (rx, tx)
            for sub, warm in ((1, 0), (1, 3), (2, 2), (3, 7), (5, 0), (16, 4), (64, 16)):
                got, _ = spec(rx, tx, sub, warm)
                assert got == want, (rx, tx, sub, warm)
