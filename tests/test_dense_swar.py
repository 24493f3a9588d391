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
