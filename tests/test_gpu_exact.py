"""GPU parity tests (-m gpu) of the exact replay (rejit_amd/csrc/exact_replay.{h,hip}): patterns at risk
of the reference's ring artefact (Q8) on texts far beyond the 1 MiB the one-lane kernel
of round 1 could take -- whole text, ranges of a sharded run, the carry-scan path, several 64-MiB batches,
a ring too big for LDS.  The expectation is always Oracle.match_all, the strict restatement of the
reference's loop, never the documented semantics."""
import random

import numpy as np
import pytest

from test_gpu_linear import gpu_spans_np, oracle_spans_np, rj, oracle  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def text_of(n, alphabet, seed):
    rng = np.random.default_rng(seed)
    return np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), size=n)].copy()


def run_ranges(rj, scan, d, n, cuts):
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        scan.run_tensor(d, own_begin=lo, own_end=hi)
        parts.append(gpu_spans_np(rj, scan).copy())
    return np.concatenate(parts) if parts else np.empty((0, 2), dtype=np.uint64)


@pytest.mark.parametrize("rx,alphabet", [(b".{0,2}.", b"abcdefghijklmnopqrstuvwxyz0123456789  \n"), (b"[a-f]+[0-9][a-f]", b"abcdef0123 "),
                                         (b"(ab|ba)+", b"abc"), (b"[xy]+z[xy]", b"xyz ")])
def test_whole_text_and_shards_8mib(rj, oracle, rx, alphabet):
    import torch
    n = 8 << 20
    t = text_of(n, alphabet, 5)
    want = oracle_spans_np(oracle, rx, t)
    spec = oracle_spans_np(oracle, rx, t, spec=True)
    d = torch.from_numpy(t).cuda()
    scan = rj.Scan(rj.Program(rx))
    cnt = scan.run_tensor(d)
    got = gpu_spans_np(rj, scan)
    st = scan.stats()
    assert cnt == len(want) and np.array_equal(got, want), (rx, cnt, len(want), len(spec))
    if len(want) != len(spec) or not np.array_equal(want, spec):
        assert st["exact_path"] == 1     # the artefact applies on this text: only the replay gets it right
    # ranges of a sharded run: every range owns whole segments between synchronisation points
    for cuts in ([0, n // 2 + 7, n + 1], [0, 1, n // 3, n // 3 + 1, 2 * n // 3 + 5, n, n + 1]):
        got = run_ranges(rj, scan, d, n, cuts)
        assert np.array_equal(got, want), (rx, cuts)


def test_artefact_is_exercised(rj, oracle):
    """`.{0,2}.` over lines: the reference and the documented semantics differ on a sizeable share of the
    matches, so the tests above do tell the two apart."""
    t = text_of(1 << 20, b"abcdefghijklmnopqrstuvwxyz0123456789  \n", 5)
    want = oracle_spans_np(oracle, b".{0,2}.", t)
    spec = oracle_spans_np(oracle, b".{0,2}.", t, spec=True)
    assert len(want) != len(spec)


def test_carry_scan_then_replay(rj, oracle, monkeypatch):
    """At-risk patterns whose candidates outlive the walk limit: the carry scan answers with the documented
    semantics, the replay then takes the reference's answer."""
    import torch
    monkeypatch.setenv("RJ_MAX_WALK", "64")
    rng = random.Random(3)
    for rx, alphabet in ((b"[xy]+z[xy]", b"x" * 60 + b"y" * 38 + b"z "),
                         (b"(aa|aaa)+", b"a" * 72 + b"bc"),
                         (b"[a-y]+z[a-y]", b"abcdefghijklmnopqrstuvwxy" * 8 + b"z ")):
        n = (4 << 20) + rng.randrange(1000)
        t = text_of(n, alphabet, rng.randrange(1 << 30))
        want = oracle_spans_np(oracle, rx, t)
        d = torch.from_numpy(t).cuda()
        scan = rj.Scan(rj.Program(rx))
        cnt = scan.run_tensor(d)
        st = scan.stats()
        assert cnt == len(want) and np.array_equal(gpu_spans_np(rj, scan), want), (rx, st)
        assert st["linear_path"] == 1 and st["exact_path"] == 1, (rx, st)


def test_thread_that_lives_for_megabytes_is_replayed_in_parts(rj, oracle):
    """`[xy]+z[xy]` inside 20 MiB of x: ONE thread holds the loop's slot from the first byte to the last, the match begins
    megabytes before it ends.  Until round 4 such a stretch was not replayed (more than 16 MiB without a synchronisation
    point).  The parts only need the ring's ORDER PATTERN at their cuts (exact_replay.h): the old thread holds its slot in a
    part's warm-up just as in the true run, under another start, and the walk puts the true one back."""
    import torch
    n = 20 << 20
    t = np.full(n, ord("x"), dtype=np.uint8)
    t[n - 6:] = np.frombuffer(b"zx xzy", dtype=np.uint8)
    t[5 << 20] = ord("z")                                  # (and one in the middle: [0, 5 MiB + 2), then a second long run)
    want = oracle_spans_np(oracle, b"[xy]+z[xy]", t)
    assert len(want) == 3 and want[0][0] == 0 and want[0][1] == (5 << 20) + 2
    d = torch.from_numpy(t).cuda()
    scan = rj.Scan(rj.Program(b"[xy]+z[xy]"))
    cnt = scan.run_tensor(d)
    assert cnt == len(want) and np.array_equal(gpu_spans_np(rj, scan), want), scan.stats()
    assert scan.stats()["exact_path"] == 2


def test_several_batches_72mib(rj, oracle):
    import torch
    n = 72 << 20
    t = text_of(n, b"abcdefghijklmnopqrstuvwxyz0123456789  \n", 77)
    t[(10 << 20):(10 << 20) + (1 << 19)] = ord("q")      # a 512 KiB stretch without a line break (taken in parts since round 4)
    want = oracle_spans_np(oracle, b".{0,2}.", t)
    d = torch.from_numpy(t).cuda()
    scan = rj.Scan(rj.Program(b".{0,2}."))
    cnt = scan.run_tensor(d)
    assert scan.stats()["exact_path"] == 2   # (1: every segment by one lane; 2: a long one in parts)
    assert cnt == len(want) and np.array_equal(gpu_spans_np(rj, scan), want)
    got = run_ranges(rj, scan, d, n, [0, (64 << 20) + 3, n + 1])
    assert np.array_equal(got, want)


def test_ring_in_global_memory(rj, oracle):
    """A long literal edge makes the ring times x states too big for LDS."""
    import torch
    rx = b"(abcdefghijklmnopqrstuvwxyzabcdefghijkl|x|xy)*z?"
    n = 2 << 20
    t = text_of(n, b"xyz", 9)
    lit = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzabcdefghijkl", dtype=np.uint8)
    for o in range(1000, n - 100, 50021):
        t[o:o + len(lit)] = lit
    want = oracle_spans_np(oracle, rx, t)
    d = torch.from_numpy(t).cuda()
    scan = rj.Scan(rj.Program(rx))
    cnt = scan.run_tensor(d)
    assert cnt == len(want) and np.array_equal(gpu_spans_np(rj, scan), want), scan.stats()
    host = rj.Program(rx).match_all(t.tobytes()[:300000])
    assert np.array_equal(np.array(host, dtype=np.uint64).reshape(-1, 2), oracle_spans_np(oracle, rx, t[:300000]))


def test_match_first_of_an_at_risk_pattern_beyond_the_first_block(rj, oracle):
    """MatchFirst / MatchAnywhere look at growing blocks of starts on a truncated buffer -- not for patterns at
    risk of the ring artefact, whose ranges own whole segments between synchronisation points: a sync-free
    stretch across the 256-KiB block boundary was owned by no round and the call reported "no match"."""
    rx = b"x{0,2}yz"
    prog = rj.Program(rx)
    assert prog.info()["ring_artefact_risk"]
    for n_x in (300000, 262143, 262144, 262146, 3 * 262144 + 5):
        text = b"x" * n_x + b"yz" + b"q" * 100
        want = oracle.match_all(rx, text)
        assert want and want[0] == (n_x - 2, n_x + 2)
        assert prog.match_first(text) == want[0], n_x
        assert prog.match_anywhere(text)
    assert prog.match_first(b"x" * 400000) is None
    assert not prog.match_anywhere(b"x" * 400000)


def test_long_stretches_in_parts(rj, oracle):
    """A stretch of megabytes without a synchronisation point (until round 4: one lane at 6 us per byte, and nothing beyond 16
    MiB): replayed in parts -- speculate and verify, exact_replay.h -- with exact_path = 2.  `.{0,2}.` over text without line
    breaks keeps a PHASE (the matches tile it in threes), so the walk needs a round per phase; the answer is the reference's
    (Oracle.match_all), which on these texts differs from the documented semantics.  20 MiB whole and as ranges of a
    sharded run, 66 MiB (the batch has to grow beyond 64 MiB), several long segments between a few line breaks, a pattern
    that can match the empty string (the sink's filter at the junction of two parts), a wider ring."""
    import torch
    risky_nullable = 0
    for rx, n, alphabet, breaks in ((b".{0,2}.", 20 << 20, b"abcde", ()), (b".{0,2}.", 66 << 20, b"ab", ()),
                                    (b".{0,2}.", 9 << 20, b"abcdefgh", (1 << 20, (1 << 20) + 1, 5 << 20, (8 << 20) + 77)),
                                    (b".{0,2}", 2 << 20, b"abc", (1 << 20,)), (b"(a|ab)(c|bcd)?(d*)", 6 << 20, b"abcd", ()),
                                    (b"[ab]{1,3}b|.{1,4}c", 5 << 20, b"abcx", (3 << 20,))):
        p = rj.Program(rx)
        if not p.info()["ring_artefact_risk"]:
            continue
        t = text_of(n, alphabet, 1234 + n)
        for b in breaks:
            t[b] = 10
        d = torch.from_numpy(t).cuda()
        scan = rj.Scan(p)
        cnt = scan.run_tensor(d)
        st = scan.stats()
        assert st["exact_path"] == 2, (rx, n, st)
        if n > (64 << 20):
            # (the oracle takes a minute over 66 MiB: its answer over the first 2 MiB -- the loop runs left to right, a match
            # that ends well inside a prefix is the same on the prefix alone -- and the shape of the rest: on text without
            # line breaks the reference's matches of `.{0,2}.` tile it, each beginning where the one before ended)
            got = gpu_spans_np(rj, scan)
            m = 2 << 20
            head = oracle_spans_np(oracle, rx, t[:m])
            k = int(np.searchsorted(head[:, 1], m - 16))
            assert np.array_equal(got[:k], head[:k]), (rx, n)
            assert cnt == len(got) and got[0, 0] == 0 and got[-1, 1] == n
            assert np.array_equal(got[1:, 0], got[:-1, 1]) and int((got[:, 1] - got[:, 0]).max()) == 3
            continue
        want = oracle_spans_np(oracle, rx, t)
        assert cnt == len(want) and np.array_equal(gpu_spans_np(rj, scan), want), (rx, n, st)
        risky_nullable += 1 if p.info()["min_len"] == 0 else 0
        if rx == b".{0,2}." and not breaks and n == 20 << 20:
            spec = oracle.match_all_spec(rx, t[: 1 << 16].tobytes())
            ref = oracle.match_all(rx, t[: 1 << 16].tobytes())
            assert spec != ref, "the reference's answer should differ from the documented semantics on this text"
            cuts = [0, 5 << 20, (5 << 20) + 1, 17 << 20, n + 1]
            assert np.array_equal(run_ranges(rj, scan, d, n, cuts), want), rx
    # (the filter at a junction only matters for patterns that match the empty string: one of them must be at risk and have run)
    assert risky_nullable >= 1 or not rj.Program(b".{0,2}").info()["ring_artefact_risk"]


def test_wide_at_risk_automaton_beyond_1mib(rj, oracle):
    """An at-risk pattern of more than 256 positions (`.{0,2}(w1|...|w40)`: until round 4 replayed by round 1's one-lane
    kernel, whole texts of at most 1 MiB, the documented semantics beyond): the synchronisation-point walk takes position
    sets of up to 1024 positions now, so the exact replay serves it on texts of any size."""
    import torch
    rng = random.Random(1)
    words = ["".join(rng.choice("abcd") for _ in range(rng.randint(6, 10))) for _ in range(40)]
    rx = b".{0,2}(" + "|".join(words).encode() + b")"
    p = rj.Program(rx)
    info = p.info()
    assert info["n_positions"] > 256 and info["ring_artefact_risk"] == 1, info
    n = 3 << 20
    t = text_of(n, b"abcd\n", 5)
    flat = t.tobytes()
    rng2 = random.Random(2)
    buf = bytearray(flat)
    for _ in range(3000):                      # plant words (some back to back, some behind one or two other bytes)
        w = rng2.choice(words).encode()
        at = rng2.randrange(0, n - 40)
        buf[at:at + len(w)] = w
        if rng2.random() < 0.5:
            w2 = rng2.choice(words).encode()
            buf[at + len(w):at + len(w) + len(w2)] = w2
    t = np.frombuffer(bytes(buf), dtype=np.uint8).copy()
    want = oracle_spans_np(oracle, rx, t)
    assert len(want) > 1000
    d = torch.from_numpy(t).cuda()
    scan = rj.Scan(p)
    cnt = scan.run_tensor(d)
    assert cnt == len(want) and np.array_equal(gpu_spans_np(rj, scan), want), scan.stats()
    assert scan.stats()["exact_path"] >= 1
