"""CPU test of the dense kernel's lane-packed pre-steps (rejit_amd/csrc/dense_swar.h): byte-parallel class
rows and four starts per register against the scalar position automaton, start by start, on random and
adversarial texts (bytes >= 0x80, range borders).  The header is compiled with g++ into the test-only driver
tests/support/carry_exec.cc; scan_dense_walk (kernels.hip) calls the same functions."""
import ctypes
import os
import random
import subprocess

import pytest

from test_carry_scan import SO, SRCS, DEPS, CSRC

DEPS = DEPS + [os.path.join(CSRC, "dense_swar.h"), os.path.join(CSRC, "exact_replay.h")]
_u64p = ctypes.POINTER(ctypes.c_uint64)


@pytest.fixture(scope="module")
def ce():
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(s) for s in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", SO] + SRCS)
    lib = ctypes.CDLL(SO)
    lib.ce_swar_check.restype = ctypes.c_long
    lib.ce_swar_check.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, _u64p]
    return lib


QUALIFY = [b"[a-f]+[0-9]", b"[a-z]+", b"[0-9]+", b"[0-9][0-9][0-9]", b"[@#]", b"[A-Z][a-z]+ [A-Z]", b"[0-9]+x", b"x[0-9]*y", b"[^a-z]",
           b"[^\\n]x", b"a.c", b"[a-cx-z0-3]+q", b"(ab|cd)+e", b"(a|b)c", b"ab?c", b"a?b", b"[\\x80-\\xff]+a", b"[^ -~]+", b"..", b"a[bc]?d?e",
           b"(x|yz)w", b"[a-f][0-9]?[a-f]?z"]
NOT = [b"^[A-Z]", b"x*", b"[a-z]+@[a-z]+abcdefghij", b"$", b"a?"]


def test_packed_presteps_equal_the_scalar_automaton(ce):
    rng = random.Random(17)
    used = walkers = decided = 0
    for rx in QUALIFY:
        for alphabet in (bytes(range(256)), b"abcdefxyz0123456789@# ABCq\n", b"ab", bytes(range(0x60, 0x90)), b"az09{`/:AZ[@"):
            tx = bytes(rng.choice(alphabet) for _ in range(16 * 400 + 4))
            st = (ctypes.c_uint64 * 4)()
            bad = ce.ce_swar_check(rx, tx, len(tx), st)
            if bad == -101:
                break
            assert bad == 0, (rx, alphabet[:8], bad)
            used += 1
            walkers += st[1]
            decided += st[2]
    assert used >= 5 * 16, used          # most of the list does qualify
    assert walkers > 1000 and decided > 1000


def test_plans(ce):
    st = (ctypes.c_uint64 * 4)()
    tx = bytes(40)
    for rx in NOT:
        assert ce.ce_swar_check(rx, tx, len(tx), st) == -101, rx
    for rx, depth in ((b"[@#]", 1), (b"[a-f][0-9]", 2), (b"[a-f]+[0-9]", 4)):
        assert ce.ce_swar_check(rx, tx, len(tx), st) == 0 and st[3] == depth, rx


def test_short_bounded_verifier_equals_the_general_walk(ce):
    """rj_lane_longest_short (device_program.h: a candidate's text in two loads, the class rows of its bytes
    fetched together) against rj_lane_longest on every start -- the nine regexdna patterns and other bounded
    patterns of up to 16 bytes, text ends included."""
    import vectors as V
    ce.ce_short_check.restype = ctypes.c_long
    ce.ce_short_check.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, _u64p]
    rng = random.Random(5)
    pats = [V.b(x["regex"]) for x in V.bench()["regexdna"]["1000"]["patterns"]]
    pats += [b"regexp", b"a", b"ab?c", b"[a-c]{2,5}x", b"(ab|cd|abcd)e?", b"a{16}", b"(a|b)(c|d)?(e|f)?", b"[acgt]{3}t|ca", b"x?y?z?", b"a{0,16}"]
    total = 0
    for rx in pats:
        for alphabet in (b"acgt", b"abcdefxyz", b"ab", b"aaab"):
            for n in (0, 1, 5, 15, 16, 17, 200):
                tx = bytes(rng.choice(alphabet) for _ in range(n))
                checked = ctypes.c_uint64(0)
                bad = ce.ce_short_check(rx, tx, n, ctypes.byref(checked))
                if bad == -101:     # (e.g. `x?`: the reference parses it as `x*`, SURVEY quirks -- no bound)
                    assert rx not in pats[:10], rx
                    continue
                assert bad == 0, (rx, tx)
                total += checked.value
    assert total > 10000
    c = ctypes.c_uint64(0)
    for rx in (b"a+", b"^ab", b"a{17}", b"[a-z]+@x"):
        assert ce.ce_short_check(rx, b"aaaa", 4, ctypes.byref(c)) == -101, rx
