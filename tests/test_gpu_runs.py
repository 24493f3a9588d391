"""GPU parity tests (-m gpu) of the run kernels (rejit_amd/csrc/run_scan.h / run_scan.hip, round 6): patterns whose match is ONE
long-lived thread in one loop position -- `X+`, `A L*`, `A L* B`, `X+ B` (reference: the NFA loop with one thread in the
loop's state, src/x64/codegen-x64.cc:535-581, last accepting position :426-461, restart behind the match :487-503) -- against
the oracle, through the C ABI:

* texts of every break density: none at all (one match of the whole text), one break per tile, a break every few bytes;
  runs that cross iteration (2 KiB), tile (8 KiB) and resolve-chunk boundaries; text sizes around those units;
* own ranges and carried-in selection state (the sharded shape), bytes >= 0x80, the end of the text as the closing break;
* 64 MiB single-run texts (what the carry scan took 4-6 ms for), bit-exact;
* patterns that look alike but do NOT have the shape keep their paths (run_path == 0).
"""
import random

import numpy as np
import pytest

from checkers import Oracle

pytestmark = pytest.mark.gpu


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


def device_text(data: bytes):
    import torch
    return torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()


SHAPES = [b"[acgt]+", b"[^>]+", b"x+", b"a[bc]*", b"a.*b", b"<[^>]*>", b"[a-f]+[0-9]", b"\"[^\"]*\"", b"a[^\\n]*z", b"[\\x80-\\xff]+", b"q[a-z]*[0-9]",
          b"[ab]+b", b"a.*a", b"[A-Z][a-z]+", b"a.+b", b"<[^>]+>", b"#.+", b"a[bc]+", b"q[a-z]+[0-9]", b"a.+a"]   # (`A L+` / `A L+ B`: lag)


def texts_for(rng, n):
    """alphabets chosen so that breaks are absent, rare or dense for the shapes above"""
    out = []
    for alphabet in (b"acgt", b"ab", b"abcxz", b"acgt>\n", b"ab<>\"q0\n", bytes(range(256)), b"aaaaaaab\n", b"xyz09af", b"ABab#<> \n"):
        out.append(bytes(rng.choice(alphabet) for _ in range(n)))
    # long runs with a break every ~10 KiB and every ~100 KiB
    for every in (10000, 100000):
        t = bytearray(rng.choice(b"acgtab") for _ in range(n))
        for p in range(rng.randrange(every), n, every):
            t[p] = rng.choice(b">\n0z\"")
        out.append(bytes(t))
    return out


def check(rj, oracle, rx, data, **kw):
    prog = rj.Program(rx)
    sc = rj.Scan(prog)
    t = device_text(data)
    n = len(data)
    got_n = sc.run(t.data_ptr(), n, **kw) if n else 0
    got = sc.spans() if n else []
    if "own_begin" in kw:
        # an independent range: the selection starts afresh at own_begin (include/rejit_hip.h: no state is carried in) -- or,
        # for the kernels that look at the byte before the range (dense_streams' run-start rule), as in the whole text: the
        # carry exchange between shards copes with either (rejit_amd/sharding.py), both are accepted here
        ob, oe = kw["own_begin"], kw.get("own_end", n + 1)
        want = [(b + ob, e + ob) for b, e in oracle.match_all(rx, data[ob:]) if b + ob < oe]
        if got != want:
            want = [m for m in oracle.match_all(rx, data) if ob <= m[0] < oe]
    else:
        want = oracle.match_all(rx, data)
    assert got == want and got_n == len(want), (rx, n, kw, got[:4], want[:4], len(got), len(want))
    return sc.stats()


def test_run_shapes_vs_oracle(rj, oracle):
    rng = random.Random(41)
    took = 0
    for n in (1, 31, 2047, 2048, 2049, 8191, 8192, 8193, 20000, 70001, 300000):
        for data in texts_for(rng, n):
            for rx in SHAPES:
                st = check(rj, oracle, rx, data)
                took += st["run_path"]
    assert took > 0, "no text took the run kernels"


def test_run_kernels_are_forced_and_exact(rj, oracle, monkeypatch):
    """RJ_RUNS_FIRST=1 (read once per process: set before the library's first run in a child) -- here the kernels are
    reached through texts dense_streams gives up (one run of the whole text) and through patterns it does not take."""
    rng = random.Random(42)
    for rx, alphabet in [(b"[acgt]+", b"acgt"), (b"a.*b", b"abcdefgh"), (b"<[^>]*>", b"<abc"), (b"[^>]+", b"acgt"), (b"x+", b"x"), (b"a[bc]*", b"abc")]:
        for n in (5000, 70000, 1 << 20):
            data = bytes(rng.choice(alphabet) for _ in range(n))
            if rx == b"a[bc]*":
                data = b"xa" + data.replace(b"a", b"b")     # (ONE run behind the only `a`)
            st = check(rj, oracle, rx, data)
            if n > 65536:   # (texts of a few KiB are match_small's: one launch for the whole MatchAll)
                assert st["run_path"] == 1, (rx, n, st)
            assert st["linear_path"] == 0


def test_run_kernels_on_break_dense_text(rj, oracle):
    """A text with a break every few bytes AND one run of 300 KB (dense_streams gives the text up, the run kernels take the
    whole of it): the iterations with breaks go through the lane-parallel form (run_scan.hip: run_iteration_par -- segments
    inside a lane's 32 bytes, segments across lanes by two prefix maxima, segments across iterations and tiles)."""
    rng = random.Random(47)
    for rx, alphabet, run in [(b"[acgt]+", b"acgtacgtacgtacgtN", b"acgt"), (b"a.*b", b"abcdefgh" * 6 + b"\n", b"acdefgh"), (b"<[^>]*>", b"<ab cd>", b"abcd "),
                              (b"[a-f]+[0-9]", b"abcdef01 ", b"abcdef"), (b"a[bc]*", b"abcbcbcbcx", b"bc"), (b"[^>]+", b"ab>", b"ab"), (b"x+", b"xy", b"x")]:
        for n in (300000, 2 << 20):
            t = bytearray(rng.choice(alphabet) for _ in range(n))
            at = rng.randrange(n // 4, n // 2)
            t[at:at + 300000] = bytes(rng.choice(run) for _ in range(min(300000, n - at)))
            if rx == b"a[bc]*":
                t[at] = ord("a")
            st = check(rj, oracle, rx, bytes(t))
            # (`<[^>]*>` is a windows-mode pattern: its candidates are the `<`, and a walk of 300 KB stays under that path's limit)
            assert (st["run_path"] == 1 or rx == b"<[^>]*>") and st["linear_path"] == 0, (rx, n, st)


def test_run_kernels_count_only(rj, oracle):
    """rj_scan_count (MatchAllCount over device text) of a run shape: the run kernels stop behind their resolve -- the count
    without the second pass; a span list is not left behind (copy_spans refuses), the next full run has one again."""
    rng = random.Random(48)
    for rx, alphabet in [(b"[acgt]+", b"acgtacgtN"), (b"a.*b", b"abcdefgh\n"), (b"x+", b"x")]:
        data = bytes(rng.choice(alphabet) for _ in range(3 << 20))
        if rx != b"x+":
            data = data[:1 << 20] + bytes(rng.choice(b"acgt") for _ in range(1 << 20)) + data[2 << 20:]   # (a run of 1 MiB: dense_streams gives up)
        want = oracle.match_all(rx, data)
        sc = rj.Scan(rj.Program(rx))
        t = device_text(data)
        assert sc.run(t.data_ptr(), len(data)) == len(want) and sc.stats()["run_path"] == 1
        assert sc.count(t.data_ptr(), len(data)) == len(want)
        st = sc.stats()
        assert st["run_path"] == 1, st
        with pytest.raises(Exception):
            sc.spans()
        assert sc.run(t.data_ptr(), len(data)) == len(want) and sc.spans() == want


def test_run_own_ranges_and_carry(rj, oracle):
    """A shard's run: begins in [own_begin, own_end) only; and the selection state carried in from the left neighbour (the
    last match before own_begin): the run kernels must neither re-open the segment a carried match with a B has closed nor
    start inside the carried match."""
    rng = random.Random(43)
    for rx, alphabet in [(b"[acgt]+", b"acgtN"), (b"a.*b", b"abcd\n"), (b"<[^>]*>", b"<ab>"), (b"[a-f]+[0-9]", b"abc19 ")]:
        data = bytes(rng.choice(alphabet) for _ in range(120000))
        n = len(data)
        full = oracle.match_all(rx, data)
        for own in [(0, 50000), (50000, n + 1), (8192, 16384), (33333, 33400), (119990, n + 1)]:
            check(rj, oracle, rx, data, own_begin=own[0], own_end=own[1])
        # with the carry: the shard's matches are the whole text's matches that begin in the range
        for cut in (40000, 65536, 99999):
            before = [m for m in full if m[0] < cut]
            prog = rj.Program(rx)
            sc = rj.Scan(prog)
            t = device_text(data)
            if before:
                b, e = before[-1]
                kw = dict(carry_cur=e if e > b else b + 1, carry_prev_end=e, have_prev=True)
            else:
                kw = {}
            k = sc.run(t.data_ptr(), n, own_begin=cut, own_end=n + 1, **kw)
            want = [m for m in full if m[0] >= cut]
            assert sc.spans() == want and k == len(want), (rx, cut, sc.spans()[:3], want[:3])


def test_run_64mib_single_run(rj, oracle):
    """`[acgt]+` over 64 MiB of acgt and `a.*b` over 64 MiB without a line break: ONE match each (the carry scan's case,
    tools/linear_probe.py), by the run kernels."""
    import torch
    dev = torch.device("cuda:0")
    n = 64 << 20
    for rx, alphabet in [(b"[acgt]+", b"acgt"), (b"a.*b", b"abcdefgh")]:
        g = torch.Generator(device="cuda").manual_seed(1)
        lut = torch.tensor(list(alphabet), dtype=torch.uint8, device=dev)
        d = lut[torch.randint(0, len(alphabet), (n,), generator=g, device=dev)].contiguous()
        host = d.cpu().numpy()
        sc = rj.Scan(rj.Program(rx))
        k = sc.run_tensor(d)
        st = sc.stats()
        if rx == b"[acgt]+":
            want = [(0, n)]
        else:
            a = int(np.argmax(host == ord("a")))
            b = n - 1 - int(np.argmax(host[::-1] == ord("b")))
            want = [(a, b + 1)]
        assert sc.spans() == want and k == 1, (rx, sc.spans(), want)
        assert st["run_path"] == 1 and st["linear_path"] == 0, st
        # a break in the middle and one near the end: three runs
        d2 = d.clone()
        d2[n // 2] = ord("\n")
        d2[n - 3] = ord("\n")
        k2 = sc.run_tensor(d2)
        h2 = d2.cpu().numpy().tobytes()
        # (the oracle takes ~1 s per 64 MiB pattern: once)
        assert sc.spans() == oracle.match_all(rx, h2) and k2 == len(sc.spans())


def test_patterns_without_the_shape_keep_their_paths(rj, oracle):
    rng = random.Random(44)
    data = bytes(rng.choice(b"abcxyz\n") for _ in range(30000))
    for rx in (b"a.*b|c", b"(ab)+", b"a+b+", b"x*", b"^a.*b", b"[ab]+c|[bc]+d", b"a.+b", b"a.*\\n", b"abc"):
        st = check(rj, oracle, rx, data)
        assert st["run_path"] == 0, (rx, st)


def test_window_mode_run_shapes_take_the_run_kernels_first(rj, oracle):
    """`a.*b`, `#.*`, `<[^>]*>`, ` +`: the fast-forward window is the ONE byte of A.  On everyday text that byte is everywhere and every hit
    is a walk (1 GiB of log-like text: 7-34 ms), so a range that reaches the text's end takes the run kernels first (0.5-1.5 ms); a run
    that found few matches sends the next one to the window scan, and a window scan that meets dense hits sends the scan object back for
    good (engine.hip: window_runs).  The answers are the oracle's on every path."""
    rng = random.Random(48)
    n = 400000
    dense = bytes(rng.choice(b"abcdefgh <>#()\n ") for _ in range(n))
    sparse = bytearray(rng.choice(b"cdefgh\n") for _ in range(n))
    # (` +` -- `X+`: dense_streams first, one pass with the runs decided in registers; the run kernels behind it)
    sc = rj.Scan(rj.Program(b" +"))
    t = device_text(dense)
    assert sc.run(t.data_ptr(), n) == len(oracle.match_all(b" +", dense)) and sc.spans() == oracle.match_all(b" +", dense)
    assert sc.stats()["stream_path"] == 1 and sc.stats()["run_path"] == 0
    for rx, plant in ((b"a.*b", b"a cd b"), (b"#.*", b"# x"), (b"<[^>]*>", b"<cd>"), (b"\\([^)]*\\)", b"(e)"), (b"#.+", b"# x")):
        sp = bytearray(sparse)
        sp[n // 2:n // 2 + len(plant)] = plant
        sp = bytes(sp)
        sc = rj.Scan(rj.Program(rx))

        def run(data):
            t = device_text(data)
            k = sc.run(t.data_ptr(), len(data))
            assert sc.spans() == oracle.match_all(rx, data) and k == len(sc.spans()), (rx, len(data))
            return sc.stats()["run_path"]
        assert run(dense) == 1, rx                 # no history: the run kernels
        assert run(dense) == 1, rx                 # dense matches: they stay
        assert run(sp) == 1, rx                    # ... and find one match in 400 KB
        assert run(sp) == 0, rx                    # few matches: the window scan (rare hits: faster)
        assert run(sp) == 0, rx
        assert run(dense) == 0, rx                 # the window scan meets dense hits once ...
        assert run(dense) == 1 and run(sp) == 1 and run(sp) == 1, rx    # ... and the scan object keeps the run kernels for good
        # an own range that ends before the text does keeps the window path (the run kernels read to the text's end)
        sc2 = rj.Scan(rj.Program(rx))
        t = device_text(dense)
        sc2.run(t.data_ptr(), n, own_begin=0, own_end=n // 2)
        assert sc2.stats()["run_path"] == 0 and sc2.spans() == [m for m in oracle.match_all(rx, dense) if m[0] < n // 2], rx


def test_line_assertions_around_run_shapes(rj, oracle):
    """`^#.*`, `#.*$`, `^a.*b`, `^[A-Z][a-z]+$`: `^` in front / `$` behind a run shape whose classes hold no line break (run_scan.h:
    RunPlan::bol / eol) -- in the run kernels `^` is a mask on the start stream (a start counts at a line start only) and `$` a test of the
    closing break (a match counts when its segment's break is a line end), so that the matches are those of the shape that begin at a line
    start / end at a line end (reference: the contexts of
    src/x64/codegen-x64.cc:686-708).  `X+` with an assertion: the second half of this test."""
    rng = random.Random(49)
    took = 0
    for n in (70001, 300000, 1 << 20):
        for alphabet in (b"ab# \n", b"abAB#<>\n\r ", b"#a\n", b"ab cd#\n" * 3 + b"AB"):
            data = bytes(rng.choice(alphabet) for _ in range(n))
            for rx in (b"^#.*", b"#.*$", b"^a.*b", b"^[AB][ab]+$", b"^#.+", b"a[b ]*$", b"^<[^>\n]*>", b"[AB][ab]+$", b"^a[b ]*$"):
                st = check(rj, oracle, rx, data)
                took += st["run_path"]
                if rx == b"^#.*":
                    check(rj, oracle, rx, data, own_begin=n // 3, own_end=n + 1)
                    sc = rj.Scan(rj.Program(rx))
                    t = device_text(data)
                    assert sc.count(t.data_ptr(), n) == len(oracle.match_all(rx, data))
    assert took > 60, took
    # `X+` between `^` / `$`: "at risk of the ring artefact" by the static analysis, but no candidate of such a pattern can begin where
    # another one ends when X holds no line break (run_scan.h) -- the run kernels, `^` as a mask on their start stream, `$` by the line
    # filter.  The oracle restates the artefact; tests/test_run_plan.py checks the rule against the real reference too.
    for n in (100000, 700001):
        data = bytes(rng.choice(b"ab# \n") for _ in range(n))
        for rx, path in ((b"[ab]+$", 1), (b" +$", 1), (b"^[ab]+$", 1), (b"^[ab]+", 1), (b"^ +$", 1)):
            st = check(rj, oracle, rx, data)
            if n > 262144:
                assert st["run_path"] == path, (rx, n, st)
    # look-alikes: `$` behind a B position, a line break inside L / X
    data = bytes(rng.choice(b"ab# \n") for _ in range(100000))
    for rx in (b"a.*b$", b"^a[^b]*", b"[^a]+$"):
        st = check(rj, oracle, rx, data)
        assert st["run_path"] == 0, (rx, st)
