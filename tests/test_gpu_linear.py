"""GPU parity tests (-m gpu) of the linear-time carry scan: inputs on which the per-start verifier
would be quadratic -- unbounded repetitions over long runs -- which round 1 answered with
RJ_TOO_LARGE where the reference (one byte per iteration, merged threads,
src/x64/codegen-x64.cc:535-640, 951-987) returns matches.

* small texts with the walk limit forced down (RJ_MAX_WALK), so every path of the carry scan --
  symbolic summaries, resolve, emit, chain selection, segments, carry -- runs against the oracle;
* 64 MiB single-line texts, bit-exact against the oracle (the strict restatement of the reference);
* the 500 MB stripped FASTA of the headline benchmark against an independent torch computation.
"""
import ctypes
import os
import random

import numpy as np
import pytest

from checkers import Oracle

pytestmark = pytest.mark.gpu

_u64p = ctypes.POINTER(ctypes.c_uint64)


@pytest.fixture(scope="module")
def rj():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import rejit_amd
    rejit_amd.build()
    rejit_amd.load_library()
    return rejit_amd


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def oracle_spans_np(oracle, rx: bytes, text: np.ndarray, spec=False):
    """oracle.match_all into a numpy array (lists of tuples do not scale to 10^7 matches)"""
    n = int(text.size)
    cap = n + 2
    out = np.empty(2 * cap, dtype=np.uint64)
    fn = oracle.lib.ro_match_all_spec_re if spec else oracle.lib.ro_match_all_re
    fn.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    cnt = fn(rx, text.ctypes.data, n, out.ctypes.data, cap)
    assert cnt >= 0, cnt
    return out[:2 * cnt].reshape(-1, 2)


def gpu_spans_np(rj, scan):
    lib = rj.load_library()
    n = int(lib.rj_scan_copy_spans(scan._h, None, 0))
    out = np.empty(2 * max(n, 1), dtype=np.uint64)
    lib.rj_scan_copy_spans(scan._h, out.ctypes.data_as(_u64p), n)
    return out[:2 * n].reshape(-1, 2)


HARD = [b"[acgt]+", b"[^>]+", b"(ab|ba)+", b"a.*b", b"x*", b"(a|b)*abb", b"[ab]+c|[bc]+d", b"^.*$", b".*x",
        b"[a-z]+@[a-z]+", b"(a|ab)(c|bcd)*", b"x*y?z+", b"[ab]{1,30}c", b"[ab]{40}c*", b"[ab]{70,90}b*", b"a{150}b*"]


def test_forced_carry_scan_vs_oracle(rj, oracle, monkeypatch):
    """Walk limit 8: almost every text takes the carry scan.  Host-text and device-text entry points,
    several sizes (one / many sub-chunks), own ranges with carry."""
    import torch
    monkeypatch.setenv("RJ_MAX_WALK", "8")
    rng = random.Random(99)
    took = took_runs = 0
    for rx in HARD:
        p = rj.Program(rx)
        sc = rj.Scan(p)
        for alphabet, n in ((b"ab", 300), (b"acgt", 5000), (b"abcx\n", 70000), (b"aabb>xyz09@.cd", 20000)):
            tx = bytes(rng.choices(alphabet, k=n))
            want = oracle.match_all(rx, tx)
            got = p.match_all(tx)
            # (the reference's answer also where its ring artefact Q8 applies: the carry scan's result is
            # then replaced by the exact replay, exact_replay.hip)
            assert got == want, (rx, alphabet, n, got[:4], want[:4])
            if n == 300:
                continue
            d = torch.frombuffer(bytearray(tx), dtype=torch.uint8).cuda()
            cnt = sc.run_tensor(d)
            took += sc.stats()["linear_path"]
            took_runs += sc.stats()["run_path"]
            assert cnt == len(want) and sc.spans() == want, (rx, alphabet, n)
            # two own ranges with the selection state carried over the cut
            cut = n // 3 + 17
            c1 = sc.run_tensor(d, own_begin=0, own_end=cut)
            first = sc.spans()
            assert c1 == len(first)
            state = dict(carry_cur=0, carry_prev_end=0, have_prev=False)
            if first:
                b, e = first[-1]
                state = dict(carry_cur=e if e > b else b + 1, carry_prev_end=e, have_prev=True)
            sc.run_tensor(d, own_begin=cut, own_end=n + 1, **state)
            assert first + sc.spans() == want, (rx, alphabet, n, cut)
    # the carry scan did run -- and (round 6) the run kernels took the texts of the patterns with ONE long-lived thread in one loop
    # position (`[acgt]+`, `[^>]+`, `a.*b`: run_scan.hip), which the carry scan answered until then
    assert took > 10 and took_runs > 0 and took + took_runs > 20, (took, took_runs)


@pytest.mark.parametrize("rx,alphabet,mib,linear", [
    (b"[acgt]+", b"acgtacgtacgtN", 64, "run"), (b"[^>]+", b"abc>", 64, "run"), (b"(ab|ba)+", b"ab", 64, False),
    (b"a.*b", b"abcdefgh", 64, "run"), (b"x*", b"xy", 64, False), (b"[a-z]+@[a-z]+", b"abcdefghij@", 8, None),
    (b"(a|b)*abb", b"abc", 8, None), (b"[ab]{40}c*", b"ab", 8, None)])
def test_long_single_line(rj, oracle, rx, alphabet, mib, linear):
    """64 MiB without a line break: candidates of tens of MiB (a.*b: ONE match over the whole text),
    runs of hundreds of KiB, 48 M empty matches, one cluster of 8 M overlapping candidates -- bit-exact
    against the oracle, whichever path the engine takes."""
    import torch
    n = mib << 20
    g = torch.Generator(device="cuda").manual_seed(1234)
    idx = torch.randint(0, len(alphabet), (n,), generator=g, device="cuda", dtype=torch.int64)
    if rx in (b"[acgt]+", b"[^>]+"):
        # long runs: thin the stop byte out to one in ~200 KiB
        stop = torch.rand(n, generator=g, device="cuda") < 5e-6
        idx = torch.where(stop, torch.full_like(idx, len(alphabet) - 1), idx % (len(alphabet) - 1))
    lut = torch.tensor(list(alphabet), dtype=torch.uint8, device="cuda")
    d = lut[idx].contiguous()
    del idx
    host = d.cpu().numpy()
    sc = rj.Scan(rj.Program(rx))
    cnt = sc.run_tensor(d)
    st = sc.stats()
    want = oracle_spans_np(oracle, rx, host)
    got = gpu_spans_np(rj, sc)
    assert cnt == len(want)
    assert np.array_equal(got, want), (rx, got[:3], want[:3])
    if linear == "run":     # (round 6: one long-lived thread in one loop position -- the run kernels, not the carry scan)
        assert st["run_path"] == 1 and st["linear_path"] == 0, st
    elif linear is not None:
        assert st["linear_path"] == int(linear), st


def test_wide_dense_pattern_64mib(rj, oracle):
    """More than 128 positions, no fast-forward window: round 1 refused more than 16 MiB per call."""
    import torch
    rx = b"[ab]{100,140}c"
    n = 64 << 20
    g = torch.Generator(device="cuda").manual_seed(7)
    r = torch.rand(n, generator=g, device="cuda")
    d = torch.where(r < 0.495, torch.full((n,), ord("a"), dtype=torch.uint8, device="cuda"),
                    torch.where(r < 0.99, torch.full((n,), ord("b"), dtype=torch.uint8, device="cuda"),
                                torch.full((n,), ord("c"), dtype=torch.uint8, device="cuda")))
    host = d.cpu().numpy()
    p = rj.Program(rx)
    assert p.info()["n_positions"] > 128 and p.info()["scan_mode"] == 0
    sc = rj.Scan(p)
    cnt = sc.run_tensor(d)
    want = oracle_spans_np(oracle, rx, host)
    assert cnt == len(want) and cnt > 1000
    assert np.array_equal(gpu_spans_np(rj, sc), want)


@pytest.mark.parametrize("rx", [b"[acgt]+", b"[^>]+"])
def test_fasta_500mb_runs(rj, rx):
    """The headline benchmark's own 500 MB text (no line breaks; its last 250 MB are ONE run of
    acgt): the matches of a one-class repetition are the maximal runs of that class, computed here
    independently with torch."""
    import torch
    from rejit_amd import workloads as W
    dev = torch.device("cuda:0")
    text = W.fasta_stripped_torch(50_000_000, dev)
    n = int(text.numel())
    cls = torch.zeros(256, dtype=torch.bool, device=dev)
    if rx == b"[acgt]+":
        cls[torch.tensor(list(b"acgt"), device=dev)] = True
    else:
        cls[:] = True
        cls[ord(">")] = False
    inside = cls[text.long()]
    prev = torch.cat([torch.zeros(1, dtype=torch.bool, device=dev), inside[:-1]])
    nxt = torch.cat([inside[1:], torch.zeros(1, dtype=torch.bool, device=dev)])
    begins = torch.nonzero(inside & ~prev).flatten()
    ends = torch.nonzero(inside & ~nxt).flatten() + 1
    del inside, prev, nxt
    sc = rj.Scan(rj.Program(rx))
    cnt = sc.run_tensor(text)
    st = sc.stats()
    assert st["run_path"] == 1 and st["linear_path"] == 0, st     # (round 6: the run kernels; until then the carry scan)
    got = torch.from_numpy(gpu_spans_np(rj, sc).astype(np.int64)).to(dev)
    assert cnt == begins.numel() and cnt > 0
    assert torch.equal(got[:, 0], begins) and torch.equal(got[:, 1], ends)
    assert int((got[:, 1] - got[:, 0]).max()) > 200_000_000   # the 250 MB run is one match


def test_wide_cyclic_automata_take_the_carry_scan(rj, oracle):
    """`(w1|...|w40)+` and `(w1|...|w110)+` (330 and 900 positions) over a hundred kilobytes of their own words: one candidate lives for
    the whole text.  Until round 4 this was the one RJ_TOO_LARGE reachable at match time (the carry scan took 256 positions);
    it takes 1024 now (16 and 32 state words)."""
    import torch
    rng = random.Random(9)
    for n_words, n_cat in ((40, 20_000), (110, 12_000)):
        words = ["".join(rng.choice("abcd") for _ in range(rng.randint(6, 10))) for _ in range(n_words)]
        rx = ("(" + "|".join(words) + ")+").encode()
        p = rj.Program(rx)
        assert p.info()["n_positions"] > 256
        text = ("".join(rng.choice(words) for _ in range(n_cat)) + "x" + "".join(rng.choice(words) for _ in range(1000)) + "ab").encode()
        t = np.frombuffer(text, dtype=np.uint8).copy()
        want = oracle_spans_np(oracle, rx, t)
        assert len(want) >= 2 and want[0][1] - want[0][0] > 90_000
        d = torch.from_numpy(t).cuda()
        scan = rj.Scan(p)
        cnt = scan.run_tensor(d)
        assert cnt == len(want) and np.array_equal(gpu_spans_np(rj, scan), want), (n_words, scan.stats())
        assert scan.stats()["linear_path"] == 1, scan.stats()


def test_automata_beyond_1024_positions_take_the_carry_scan(rj, oracle, monkeypatch):
    """`(w1|...|w250)+`, `(w1|...|w480)+`, `(w1|...|w900)+` (2000, 3900 and 7200 positions: 64, 128 and 256 state words) over their own
    words with a walk limit of 4096: one candidate lives for the whole text and the run goes to the carry scan.  Until round 5
    this was RJ_TOO_LARGE at match time; the reference has no such error.  (Without the limit such a pattern
    walks a start for up to 2^20 bytes first: the carry scan's cost grows with the square of the width.)"""
    import torch
    monkeypatch.setenv("RJ_MAX_WALK", "4096")
    rng = random.Random(21)
    for n_words, n_cat in ((250, 2000), (480, 800), (900, 600)):
        words = ["".join(rng.choice("abcd") for _ in range(rng.randint(7, 9))) for _ in range(n_words)]
        rx = ("(" + "|".join(words) + ")+").encode()
        p = rj.Program(rx)
        assert p.info()["n_positions"] > 1024 * (1 if n_words == 250 else 2 if n_words == 480 else 4)
        text = ("".join(rng.choice(words) for _ in range(n_cat)) + "x" + "".join(rng.choice(words) for _ in range(300)) + "ab").encode()
        t = np.frombuffer(text, dtype=np.uint8).copy()
        want = oracle_spans_np(oracle, rx, t)
        assert len(want) >= 2 and want[0][1] - want[0][0] > 4500
        d = torch.from_numpy(t).cuda()
        scan = rj.Scan(p)
        cnt = scan.run_tensor(d)
        assert cnt == len(want) and np.array_equal(gpu_spans_np(rj, scan), want), (n_words, scan.stats())
        assert scan.stats()["linear_path"] == 1, scan.stats()
