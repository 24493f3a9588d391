"""GPU parity (-m gpu) of the bit-stream dense kernel (rejit_amd/csrc/dense_streams.hip; algorithm CPU-tested in
tests/test_dense_streams.py) through the C ABI against the oracle: every shape of the plan (1..8 positions, 1..8 ranges,
loops, alternations of chains, high-half classes), text sizes around the lane / iteration / tile edges (32 B, 2 KiB,
32 KiB, the 16-byte start frame), texts made of matches only (the staged tile overflows and is written directly),
runs longer than the register steps (the scalar walk), own ranges of a sharded run, and the fallbacks (a run longer than
max_walk -> scan_dense_walk -> carry scan).  Replaces for these patterns the reference's no-fast-forward loop,
src/x64/codegen-x64.cc:535-677."""
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


def run_scan(rj, scan, data: bytes, **kw):
    import torch
    t = torch.from_numpy(np.frombuffer(data + b"\0" * 16, dtype=np.uint8).copy()).cuda()
    c = scan.run(t.data_ptr(), len(data), **kw)
    spans = scan.spans()
    assert c == len(spans)
    return spans, scan.stats()


SHAPES = [(b"[a-f]+[0-9]", b"abcdefgz0123 \n"), (b"[@#]", b"ab@#c"), (b"[a-h][i-p]", b"abcdefghijklmnop"), (b"[a-h]+[i-p]", b"abcdefghijklmnop"),
          (b"[a-p]", b"abcdefghijklmnop"), (b"[a-p]+", b"abcdefghijklmnopz"), (b"[0-9]+x", b"0123x y"), (b"[^a-z]", b"abc, d.E"),
          (b"[a-cx-z0-3]+q", b"abcxyz0123q "), (b"[\x80-\xff]+a", bytes(range(0x78, 0x88)) + b"a"), (b"[a-f]+[0-9]+[g-k]", b"abc012ghz"),
          (b"abc[0-9]", b"abc012"), (b"(ab|cd)", b"abcd"), (b"[ab]c|[de]f|g", b"abcdefg"), (b"[a-f]+[0-9][^a-f0-9]", b"abc012xyz"),
          (b"x[0-9]+y", b"x01y"), (b"[a-b][c-d][e-f][g-h][i-j][k-l][m-n]", b"abcdefghijklmn"), (b"[a-b][c-d][e-f][g-h][i-j][k-l][m-n][o-p]", b"abcdefghijklmnop")]


def test_stream_kernel_shapes_vs_oracle(rj, oracle):
    rng = random.Random(3)
    took = {}
    for rx, alphabet in SHAPES:
        p = rj.Program(rx)
        scan = rj.Scan(p)
        for n in (1, 15, 16, 17, 31, 32, 33, 2047, 2048, 2049, 2063, 2064, 2065, 32767, 32768, 32769, 32783, 32784, 32785, 70001, 300000):
            text = bytes(rng.choice(alphabet) for _ in range(n))
            got, st = run_scan(rj, scan, text)
            assert got == oracle.match_all(rx, text), (rx, n)
            took[rx] = took.get(rx, 0) + st["stream_path"]
    # (texts of <= 16 KB take match_small; patterns with a literal window are not dense; `[a-p]+` over a..p is one long run)
    for rx in (b"[a-f]+[0-9]", b"[@#]", b"[a-h][i-p]", b"[a-h]+[i-p]", b"[a-p]", b"[^a-z]", b"[a-f]+[0-9]+[g-k]",
               b"[a-b][c-d][e-f][g-h][i-j][k-l][m-n]", b"[a-b][c-d][e-f][g-h][i-j][k-l][m-n][o-p]"):
        assert took[rx] >= 8, (rx, took)


def test_stream_kernel_random_ascii_and_ranges(rj, oracle):
    """The bench workload's shape at a size the oracle finishes (random ASCII, `[a-f]+[0-9]`), whole and as the own ranges of a
    sharded run: a match belongs to the range that holds its begin, the run-start rule looks at the byte before the range."""
    from rejit_amd import workloads as W
    n = 3 << 20
    text = W.random_ascii_numpy(n, seed=99).tobytes()
    for rx in (b"[a-f]+[0-9]", b"[@#]", b"[a-z]+[0-9]", b"[0-9]+x"):
        p = rj.Program(rx)
        scan = rj.Scan(p)
        want = oracle.match_all(rx, text)
        got, st = run_scan(rj, scan, text)
        assert got == want, rx
        # (`[0-9]+x` has a window behind its prefix; as an `X+ B` run shape over a text where that window -- one byte -- is hit every 74
        # bytes it takes this kernel first too since the last session of round 6: engine.hip, window_runs)
        assert st["stream_path"] == 1, rx
        for lo, hi in ((0, n // 3), (n // 3, n // 2 + 5), (n // 2 + 5, n + 1), (32768 - 16, 32768 + 16), (65536 - 15, 65536 - 14)):
            got, st = run_scan(rj, scan, text, own_begin=lo, own_end=hi)
            assert got == [m for m in want if lo <= m[0] < hi], (rx, lo, hi)


def test_stream_kernel_long_runs_fall_back(rj, oracle):
    """Runs longer than the 16 register steps take the scalar walk; runs longer than max_walk void the run, which
    scan_dense_walk and then the carry scan repeat -- same answers, and the scan object remembers."""
    rng = random.Random(5)
    parts = []
    for _ in range(3000):
        parts.append(bytes(rng.choice(b"abcdef") for _ in range(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 70, 200]))))
        parts.append(rng.choice([b"1", b"5 ", b"x", b" ", b"9z", b"0123x"]))
    text = b"".join(parts)
    p = rj.Program(b"[a-f]+[0-9]")
    scan = rj.Scan(p)
    want = oracle.match_all(b"[a-f]+[0-9]", text)
    for call in range(3):
        got, st = run_scan(rj, scan, text)
        assert got == want, call
        if call == 0:
            assert st["stream_path"] == 1 and st["slow_starts"] > 100
    long_text = b"q" + b"abcdef" * 20000 + b"7 " + text
    want = oracle.match_all(b"[a-f]+[0-9]", long_text)
    scan2 = rj.Scan(p)
    for call in range(2):
        got, st = run_scan(rj, scan2, long_text)
        assert got == want, call
        assert st["stream_path"] == 0


def test_prefix_scan_across_unit_groups(rj, oracle):
    """The single-write kernels place their pairs with a prefix scan over UNITS of 128 KiB (a workgroup's four tiles) that
    form GROUPS of 64 (8 MiB), resolved a round late (rejit_amd/csrc/tile_lookback.h): texts around one, two and five
    groups -- a last group that is partial, complete, or one unit long --, whole and as own ranges that begin inside a
    group, for the bit-stream kernel (`[@#]`, `[a-f]+[0-9]`) and the line table (`^`, `$`), against the oracle."""
    from rejit_amd import workloads as W
    unit, group = 128 << 10, 8 << 20
    base = W.random_ascii_numpy(5 * group + 3 * unit + 777, seed=123)
    base[np.arange(60, base.size, 61)] = 10       # a line break every 61 bytes, like the bench's line table
    base[np.arange(7000, base.size, 9973)] = 13
    for n in (group - 1024, group, group + unit, group + unit + 1, 2 * group + 5, base.size):
        text = base[:n].tobytes()
        for rx in (b"[@#]", b"[a-f]+[0-9]", b"^", b"$"):
            if n > 3 * group and rx in (b"[a-f]+[0-9]", b"$"):
                continue                               # (the largest text once per kernel)
            p = rj.Program(rx)
            scan = rj.Scan(p)
            want = oracle.match_all(rx, text)
            got, st = run_scan(rj, scan, text)
            assert got == want, (rx, n)
            if rx in (b"[@#]", b"[a-f]+[0-9]"):
                assert st["stream_path"] == 1, (rx, n)
            if n == 2 * group + 5:
                for lo, hi in ((group - 7, group + 3 * unit + 1), (group + 5 * unit + 17, n + 1), (3 * unit, 2 * group - unit)):
                    got, st = run_scan(rj, scan, text, own_begin=lo, own_end=hi)
                    assert got == [m for m in want if lo <= m[0] < hi], (rx, n, lo, hi)


SELECT_SHAPES = [b"[0-9][0-9][0-9]", b"..", b"a.c", b"[a-c]{4}d", b"(ab|ba)", b"[a-z][a-z][0-9]", b"[ab][bc][cd]", b"[0-9]{8}", b"aa|aab", b"[ab]{2,3}"]


def test_selection_in_the_stream_kernel_vs_oracle(rj, oracle):
    """Candidates that CAN overlap (`[0-9][0-9][0-9]`): since round 5 the bit-stream kernel makes the
    reference's left-most-longest selection itself (StreamPlan::select, rj_stream_select; src/codegen.cc:36-86,
    src/x64/codegen-x64.cc:401-466).  Sizes around the lane / iteration / tile edges; sparse text (the kernel answers: every
    tile finds a lane without a match before it), text packed with matches beyond one tile (the run is void and repeated on
    scan_dense_walk: the answer is still the oracle's), a packed stretch across a tile edge inside sparse text, own ranges."""
    rng = random.Random(61)
    sparse_alphabet = bytes(range(0x30, 0x7a))
    took = void = direct = 0
    for rx in SELECT_SHAPES:
        p = rj.Program(rx)
        for n in (17, 33, 2047, 2049, 20000, 32767, 32769, 34817, 70001, 300000, 1 << 20):
            scan = rj.Scan(p)   # (a void run switches the stream kernel off for its scan object)
            text = bytearray(rng.choice(sparse_alphabet) for _ in range(min(n, 200000)))
            text = (text * (n // len(text) + 1))[:n]
            for edge in range(32768, n - 100, 32768):
                text[edge - 300:edge - 40] = bytes(rng.choice(b"0123456789ab") for _ in range(260))
                text[edge - 9:edge + 9] = b"12ab34ba5678901234"
            text = bytes(text)
            got, st = run_scan(rj, scan, text)
            assert got == oracle.match_all(rx, text), (rx, n, st)
            took += st["stream_path"]
            if n == 70001:
                want = got
                cut = 32768 + 5
                first, _ = run_scan(rj, scan, text, own_begin=0, own_end=cut)
                assert first == [m for m in want if m[0] < cut], (rx, cut)
                # a range that begins inside the second tile: the selection restarts there (nothing carried in)
                lo = 40000
                rest, _ = run_scan(rj, scan, text, own_begin=lo, own_end=n + 1)
                assert rest == [(b + lo, e + lo) for b, e in oracle.match_all(rx, text[lo:])], (rx, lo)
        # ONE tile packed with matches: more pairs than a stage holds, so the tile is computed a second time writing directly --
        # with the selection made again (no tile before it: the kernel answers)
        scan = rj.Scan(p)
        one_tile = bytes(rng.choice(b"0123456789ab") for _ in range(30000))
        got, st = run_scan(rj, scan, one_tile)
        assert got == oracle.match_all(rx, one_tile), (rx, "one packed tile")
        direct += st["stream_path"]
        scan = rj.Scan(p)
        packed = bytes(rng.choice(b"0123456789ab") for _ in range(100000))
        got, st = run_scan(rj, scan, packed)
        assert got == oracle.match_all(rx, packed), (rx, "packed")
        void += 1 - st["stream_path"]
    assert took >= 30 and void >= 5 and direct >= 5, (took, void, direct)
