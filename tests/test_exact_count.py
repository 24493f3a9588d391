"""CPU test of the table behind MatchAllCount-in-one-kernel (rejit_amd/csrc/exact_count.h, used by plane_count.hip):
make_exact_count_plan builds, from the lowered automata, the table `tab[base][mismatch position][byte class]` -> patterns, and
refuses every set whose languages are not exactly "one byte off a base window"; exact_classify is the code a lane of the
kernel runs on a candidate's eight bytes.  Checked against the oracle's MatchFull on the 2 x 8 x 256 one-off strings, on
strings two bytes off, and on random bytes (incl. >= 0x80)."""
import ctypes
import os
import random
import subprocess

import pytest

from checkers import Oracle
from test_lowering import SO, SRCS, ROOT

REGEXDNA = [b"agggtaaa|tttaccct", b"[cgt]gggtaaa|tttaccc[acg]", b"a[act]ggtaaa|tttacc[agt]t", b"ag[act]gtaaa|tttac[agt]ct",
            b"agg[act]taaa|ttta[agt]cct", b"aggg[acg]aaa|ttt[cgt]ccct", b"agggt[cgt]aa|tt[acg]accct", b"agggta[cgt]a|t[acg]taccct",
            b"agggtaa[cgt]|[acg]ttaccct"]
BASES = [b"agggtaaa", b"tttaccct"]
TAB_WORDS = 64 + 2 * 9 * 16


@pytest.fixture(scope="module")
def pe():
    deps = SRCS + [os.path.join(ROOT, "rejit_amd", "csrc", h) for h in ("lowering.h", "exact_count.h", "table_layout.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(SO) < os.path.getmtime(s) for s in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", SO] + SRCS)
    lib = ctypes.CDLL(SO)
    lib.pe_exact_plan.restype = ctypes.c_int
    lib.pe_exact_plan.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    lib.pe_exact_classify.restype = ctypes.c_uint32
    lib.pe_exact_classify.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_char_p]
    return lib


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def plan(pe, rxs, bases):
    arr = (ctypes.c_char_p * len(rxs))(*rxs)
    table = (ctypes.c_uint32 * TAB_WORDS)()
    base = (ctypes.c_uint32 * 4)()
    rc = pe.pe_exact_plan(arr, len(rxs), b"".join(bases), len(bases), table, base)
    return rc, table, base


def want_mask(oracle, rxs, s):
    m = 0
    for p, rx in enumerate(rxs):
        if oracle.match_full(rx, s):
            m |= 1 << p
    return m


def test_regexdna_table_equals_the_oracle(pe, oracle):
    rc, table, base = plan(pe, REGEXDNA, BASES)
    assert rc == 1
    rng = random.Random(5)
    cases = []
    for b in BASES:
        cases.append(b)
        for j in range(8):
            for c in range(256):
                cases.append(b[:j] + bytes([c]) + b[j + 1:])
        for _ in range(300):     # two bytes off: nothing may match
            j, k = rng.sample(range(8), 2)
            s = bytearray(b)
            s[j] = rng.choice(b"acgtN\x00\xff\xe1")
            s[k] = rng.choice(b"acgtn\x80\x7f")
            cases.append(bytes(s))
    for _ in range(2000):
        cases.append(bytes(rng.choice(b"acgt") for _ in range(8)))
        cases.append(bytes(rng.randrange(256) for _ in range(8)))
    bad = []
    for s in cases:
        got = pe.pe_exact_classify(table, base, 2, s)
        want = want_mask(oracle, REGEXDNA, s)
        if got != want:
            bad.append((s, got, want))
    assert not bad, bad[:5]


def test_other_sets_taken(pe, oracle):
    # one base only; classes with high bytes and a negated class at the free position; a literal one byte off the base
    rxs = [b"abcdefgh", b"abc[\x80-\xff]efgh", b"abcdef[^g]h", b"xbcdefgh"]
    rc, table, base = plan(pe, rxs, [b"abcdefgh"])
    assert rc == 1
    rng = random.Random(9)
    for _ in range(4000):
        s = bytearray(b"abcdefgh")
        for _ in range(rng.choice([0, 1, 1, 1, 2])):
            s[rng.randrange(8)] = rng.choice([rng.randrange(256), ord("g"), ord("x"), 0x80, 0xff, 0x7f])
        s = bytes(s)
        assert pe.pe_exact_classify(table, base, 1, s) == want_mask(oracle, rxs, s), s


def test_sets_refused(pe):
    # a language with strings further than one byte from the bases; a shorter / longer / unbounded pattern; assertions;
    # two bases fewer than three bytes apart; more byte classes than the table holds
    assert plan(pe, [b"a[cg][cg]gtaaa"], [b"agggtaaa"])[0] == 0
    assert plan(pe, [b"agggtaaa|tttacccta"], BASES)[0] == 0
    assert plan(pe, [b"agggtaa"], [b"agggtaaa"])[0] == 0
    assert plan(pe, [b"agggtaaa+"], [b"agggtaaa"])[0] == 0
    assert plan(pe, [b"^agggtaaa"], [b"agggtaaa"])[0] == 0
    assert plan(pe, [b"agggtaaa|agggtacc"], [b"agggtaaa", b"agggtacc"])[0] == 0
    many = [b"abcdefg[" + bytes([c]) + b"]" for c in b"0123456789ABCDEFGHIJ"]
    assert plan(pe, many, [b"abcdefgh"])[0] == 0
    # ... and the same shape within the limits is taken
    assert plan(pe, many[:10], [b"abcdefgh"])[0] == 1
