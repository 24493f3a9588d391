"""GPU parity tests (-m gpu) of the windows BEHIND an unbounded prefix: `.*regexp`, `[a-z]+abcdefgh`,
`\\d+regexp`, ... -- patterns the reference fast-forwards on through any literal, running its NFA
backwards from the hit (src/codegen.cc:352-383, src/x64/codegen-x64.cc:643-650) and round 1 walked from
every start (dense mode, 0.6 TB/s).  Here: fast-forward window scan + verify_behind_in_regions
(behind_walk.h), with the dense path / carry scan behind it when candidates conflict or walks run long.
"""
import random

import numpy as np
import pytest

from checkers import Oracle
from test_gpu_linear import gpu_spans_np, oracle_spans_np

pytestmark = pytest.mark.gpu

PATS = [b".*regexp", b"[a-z]+abcdefgh", b"\\d+regexp", b"[0-9]+x", b"[A-Z][a-z]+ [A-Z][a-z]+", b"[ab]*abb", b"^.*foo", b"[a-z]+@[a-z]+",
        b".*ab.*cd", b"a+(bc|bd)e*", b"[ab]+(c|dd)+x", b"[ab]{30,}cd", b"([ab]{3}c){12,}xy", b"([complex]|(regexp)){2,}abcdefgh(at|the)"]


@pytest.fixture(scope="module")
def rj():
    import torch
    assert torch.cuda.is_available()
    import rejit_amd
    rejit_amd.build()
    return rejit_amd


def test_plans_are_windows(rj):
    for rx in (b".*regexp", b"[a-z]+abcdefgh", b"\\d+regexp", b"([complex]|(regexp)){2,}abcdefgh(at|the)"):
        info = rj.Program(rx).info()
        assert info["scan_mode"] == 1 and info["n_windows"] >= 1, (rx, info)


def test_small_texts_vs_oracle(rj):
    oracle = Oracle()
    rng = random.Random(31)
    n_cases = 0
    for rx in PATS:
        p = rj.Program(rx)
        for alphabet in (b"abregxp0\n", b"ab", b"abcdx \nAB@", b"abcdefgh12x"):
            for n in (9, 300, 6000):
                tx = bytes(rng.choices(alphabet, k=n))
                for plant in (b"", b"regexp", b"abcdefgh"):
                    t2 = tx[:n // 2] + plant + tx[n // 2:] + plant
                    want = oracle.match_all(rx, t2)
                    spec = oracle.match_all_spec(rx, t2)
                    got = p.match_all(t2)
                    assert got in (spec, want), (rx, alphabet, n, plant, got[:3], spec[:3])
                    n_cases += want == spec
    assert n_cases > 300


@pytest.mark.parametrize("rx,lo,hi,plant", [(b".*regexp", "0", "z", b"regexp"), (b"[a-z]+abcdefgh", "0", "z", b"qabcdefgh"),
                                              (b"\\d+regexp", "0", "z", b"7regexp"), (b"[0-9]+x", "0", "z", b"7x"),
                                              (b"([complex]|(regexp)){2,}abcdefgh(at|the)", "0", "z", b"complexabcdefghthe")])
def test_large_random_ascii_vs_oracle(rj, rx, lo, hi, plant):
    """256 MiB of the benchmark harness's random text (line-free for `.*`: ONE candidate start per hit at
    the very beginning of the text), planted occurrences; bit-exact against the oracle on a 32 MiB slice
    and planted-occurrence recovery on the whole text."""
    import torch
    from rejit_amd import workloads as W
    oracle = Oracle()
    dev = torch.device("cuda:0")
    n = 256 << 20
    t = W.random_ascii_torch(n, 77, dev, ord(lo), ord(hi))
    if rx == b".*regexp":
        nl = torch.arange(100, n, 997, device=dev)     # `.` stops at a line break: lines of ~1 KB
        t[nl] = 10
    offs = W.plant_offsets(n, len(plant), 2000, seed=9, boundaries=[1024, 1 << 20, 32 << 20])
    W.plant(t, offs, plant)
    sc = rj.Scan(rj.Program(rx))
    m = 32 << 20
    cnt = sc.run_tensor(t[:m].contiguous())
    want = oracle_spans_np(oracle, rx, t[:m].cpu().numpy())
    assert cnt == len(want) and np.array_equal(gpu_spans_np(rj, sc), want), rx
    cnt = sc.run_tensor(t)
    st = sc.stats()
    spans = gpu_spans_np(rj, sc)
    ends = set(int(e) for e in spans[:, 1])
    begins = spans[:, 0]
    assert np.all(begins[1:] >= spans[:-1, 1])          # ordered, disjoint
    covered = 0
    for o in offs:                                      # every planted occurrence lies inside a match
        i = int(np.searchsorted(begins, o, side="right")) - 1
        covered += i >= 0 and int(spans[i, 1]) >= o + len(plant)
    assert covered == len(offs), (rx, covered, len(offs), st)


def test_many_hits_per_region_matches_before_line_breaks(rj):
    """Dozens of window hits in one region (every lane of the verify wave busy, walks of very different lengths) with
    matches that end right before a line break or at the end of the text: the case in which a run-time "has
    contexts" flag, kept by the compiler as a lane mask of the lanes still inside ONE walk's loop, sent the longest
    walkers of the NEXT walk down the context path of an automaton without contexts (rejit_amd/csrc/lds_walk.h: CTX is a
    template parameter since).  1 text in 750 showed it."""
    oracle = Oracle()
    rng = random.Random(5)
    for rx, plant in ((b"[a-z]+@[a-z]+", b"Ad@abcdefgh"), (b"[a-z]+abcdefgh", b"0qqabcdefgh"), (b"\\d+x[a-f]+", b"A12xabcdefabc")):
        p = rj.Program(rx)
        for trial in range(700):
            n = rng.randrange(20, 700)
            alphabet = b"abcdx \nAB@" if trial % 2 else b"ABCDEFGH0123 x\n"
            t2 = bytes(rng.choices(alphabet, k=n)) + plant
            got = p.match_all(t2)
            assert got in (oracle.match_all_spec(rx, t2), oracle.match_all(rx, t2)), (rx, trial, len(t2))
