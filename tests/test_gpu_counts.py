"""GPU tests (-m gpu) of MatchAllCount in one kernel (rj_multi_set_counts_only; rejit_amd/csrc/plane_count.hip +
exact_count.h; reference src/rejit.cc:203-208, sample/regexdna.cc:65): through the C ABI, against the oracle and against
the span pipeline of the same object --

* the nine regexdna patterns over FASTA texts and over random bytes (all 256 values) with planted matches, whole texts
  and own ranges, at sizes around the kernel's 2-KiB blocks and the end-of-text path;
* matches that straddle block, span and text ends; the first / last match per pattern (rj_multi_bounds) that the carry
  exchange between shards needs, equal to the span pipeline's;
* texts on which the count is NOT the number of matching positions (two matches of one pattern fewer than 8 bytes apart):
  an isolated pair inside a span is resolved by the kernel, a chain of three / a pair across two waves' spans / a block
  that overfills a wave's ring make THAT run void and the span pipeline answer (return value 1), with the oracle's counts;
* the reference's own entry point: MatchAllCount of ONE pattern (rj_match_all(..., NULL), rj_scan_count);
* start / finish with two objects in flight; the sharded count (virtual shards + carry exchange) in counts mode.
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


@pytest.fixture(scope="module")
def W():
    from rejit_amd import workloads
    return workloads


def device_text(data: bytes):
    import torch
    t = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    # (torch allocations are 256-byte aligned: the library asks for 16)
    return t


def both_ways(rj, progs, data, own=None):
    """(counts-only counts, how, span counts, counts-mode bounds, span bounds) of one text."""
    t = device_text(data)
    kw = {} if own is None else {"own_begin": own[0], "own_end": own[1]}
    mc = rj.MultiScan(progs)
    assert mc.set_counts_only(True)
    c = mc.run(t.data_ptr(), len(data), **kw)
    how = mc.how
    bc = mc.bounds()
    ms = rj.MultiScan(progs)
    s = ms.run(t.data_ptr(), len(data), **kw)
    assert ms.how == 1 or (own is not None and own[0] >= own[1])
    bs = ms.bounds()
    return c, how, s, bc, bs


def oracle_counts(oracle, patterns, data, own=None):
    out = []
    for rx in patterns:
        sp = oracle.match_all(rx.encode() if isinstance(rx, str) else rx, data)
        if own is not None:
            sp = [m for m in sp if own[0] <= m[0] < own[1]]
        out.append(len(sp))
    return out


def test_counts_equal_oracle_on_fasta(rj, oracle, W):
    progs = [rj.Program(rx) for rx in W.REGEXDNA_PATTERNS]
    for n_fa in (300, 5000, 50000):
        data = W.fasta_stripped_numpy(n_fa).tobytes()
        c, how, s, bc, bs = both_ways(rj, progs, data)
        assert how == 3, "the regexdna set must take the one-kernel path"
        want = oracle_counts(oracle, W.REGEXDNA_PATTERNS, data)
        assert c == want and s == want, (n_fa, c, s, want)
        assert bc == bs, (n_fa, bc, bs)


def planted_text(rng, n, alphabet, strings, density):
    t = bytearray(rng.choice(alphabet) for _ in range(n))
    k = max(1, int(n * density))
    for _ in range(k):
        s = rng.choice(strings)
        if n >= len(s):
            p = rng.randrange(0, n - len(s) + 1)
            t[p:p + len(s)] = s
    return t


def spaced_text(rng, n, alphabet, strings, every=64):
    """planted strings at least 24 bytes apart (no two matches of any pattern overlap), on a background that cannot match"""
    t = bytearray(rng.choice(alphabet) for _ in range(n))
    for slot in range(0, n - every, every):
        if rng.random() < 0.5:
            p = slot + rng.randrange(0, every - 32)
            s = rng.choice(strings)
            t[p:p + len(s)] = s
    return t


def regexdna_strings():
    out = []
    for b in (b"agggtaaa", b"tttaccct"):
        out.append(b)
        for j in range(8):
            for c in b"acgt":
                out.append(b[:j] + bytes([c]) + b[j + 1:])
    return out


def test_counts_at_block_and_text_edges(rj, oracle, W):
    """Sizes around the 2-KiB blocks (the guarded end-of-text path takes the last one or two), matches planted across
    every block boundary and ending exactly at the end of the text; random bytes of all 256 values around them."""
    progs = [rj.Program(rx) for rx in W.REGEXDNA_PATTERNS]
    strings = regexdna_strings()
    rng = random.Random(11)
    sizes = [16, 17, 23, 24, 100, 2040, 2047, 2048, 2049, 2056, 4095, 4096, 4097, 4103, 6144, 6150, 10000, 65536, 65543, 200001]
    for n in sizes:
        for alphabet in (b"acgt", bytes(range(256))):
            t = planted_text(rng, n, alphabet, strings, 0.004)
            for edge in range(2048, n + 8, 2048):   # across every block boundary, at every offset
                s = rng.choice(strings)
                p = edge - rng.randrange(1, 8)
                if 0 <= p and p + 8 <= n:
                    t[p:p + 8] = s
            if n >= 8:
                t[n - 8:n] = rng.choice(strings)
                t[0:8] = rng.choice(strings)
            data = bytes(t)
            want = oracle_counts(oracle, W.REGEXDNA_PATTERNS, data)
            c, how, s, bc, bs = both_ways(rj, progs, data)
            assert s == want, (n, s, want)
            assert c == want, (n, how, c, want)
            assert bc == bs, (n, how, bc, bs)


def test_counts_of_own_ranges(rj, oracle, W):
    """A shard's run: match begins in [own_begin, own_end) only, the halo inside the buffer."""
    progs = [rj.Program(rx) for rx in W.REGEXDNA_PATTERNS]
    rng = random.Random(12)
    data = bytes(planted_text(rng, 150000, b"acgt", regexdna_strings(), 0.003))
    for own in [(0, 70000), (70000, 150001), (2048, 4096), (4090, 4100), (12345, 99999), (149990, 150001), (0, 5), (5, 5)]:
        want = oracle_counts(oracle, W.REGEXDNA_PATTERNS, data, own)
        c, how, s, bc, bs = both_ways(rj, progs, data, own)
        assert c == want and s == want, (own, how, c, s, want)
        assert bc == bs, (own, bc, bs)


def test_overlapping_pairs_are_resolved_in_the_kernel(rj, oracle, W):
    """`agggtaaagggtaaa`: two matches of pattern 0 seven bytes apart -- the reference selects the first only
    (src/x64/codegen-x64.cc:401-466, :448-460), the number of matching positions is one more.  Round 6: an isolated pair
    inside a wave's span is resolved by the kernel itself (return value 3, the oracle's counts), at any offset to the
    2-KiB blocks and across the 64-candidate batches; pairs of DIFFERENT patterns that close both count."""
    progs = [rj.Program(rx) for rx in W.REGEXDNA_PATTERNS]
    rng = random.Random(13)
    # (5 MB: 2442 blocks over the 1024 waves of a 256-workgroup grid -- the first waves own three blocks each, so the block
    # boundaries at 2048 and 4096 lie INSIDE wave 0's span and the one at 6144 between two spans.  Texts below 2 MiB give
    # every wave one block: every block boundary is a span boundary there, see the next test.)
    n = 5000000
    base = bytearray(np.random.default_rng(13).choice(np.frombuffer(b"acgt", dtype=np.uint8), n).tobytes())
    overlap = b"agggtaaagggtaaa"
    for at in [1000, 2048 - 7, 2048 - 3, 4096 - 1, 4096 - 5, 3000000, n - 15]:
        t = bytearray(base)
        t[at:at + len(overlap)] = overlap
        data = bytes(t)
        want = oracle_counts(oracle, W.REGEXDNA_PATTERNS, data)
        c, how, s, bc, bs = both_ways(rj, progs, data)
        assert c == want and s == want, (at, how, c, want)
        assert how == 3, (at, "an isolated overlapping pair inside a span is the kernel's own business")
        assert bc == bs, (at, bc, bs)
    base = base[:400000]
    # many isolated pairs of random one-off strings at every distance 1..10, 40 bytes apart, on a background that cannot match
    strings = regexdna_strings()
    t = bytearray(b"c" * 300000)
    at = 100
    while at + 64 < len(t):
        a, b2, d = rng.choice(strings), rng.choice(strings), rng.randrange(1, 11)
        t[at:at + 8] = a
        if rng.random() < 0.5:
            t[at + d:at + d + 8] = b2          # (overwrites the tail of the first: whatever matches then, the oracle decides)
        else:
            keep = bytes(t[at:at + 8])
            t[at + d:at + d + 8] = b2
            t[at:at + 8] = keep                # ... or the head of the second
        at += 40 + rng.randrange(0, 30)
    data = bytes(t)
    want = oracle_counts(oracle, W.REGEXDNA_PATTERNS, data)
    c, how, s, bc, bs = both_ways(rj, progs, data)
    assert c == want and s == want, (how, c, s, want)
    assert bc == bs
    # a pair 8 bytes apart does not overlap
    t = bytearray(base)
    t[5000:5016] = b"agggtaaaagggtaaa"
    data = bytes(t)
    c, how, s, bc, bs = both_ways(rj, progs, data)
    assert c == oracle_counts(oracle, W.REGEXDNA_PATTERNS, data) and how == 3


def test_chains_are_walked_and_cut_pairs_void_THAT_run_only(rj, oracle, W):
    """Three matches in a row, each inside the one before (`agggtaaagggtaaagggtaaa`: the reference keeps the first and the
    third): the kernel walks the pattern's matches itself (return value 3).  A pair that lies across two waves' spans makes
    the kernel flag the run and the span pipeline answer (return value 1, the oracle's counts) -- and the NEXT run of the
    same object, on a clean text, is back on the counts path (round 5 left the object on the span pipeline for good)."""
    progs = [rj.Program(rx) for rx in W.REGEXDNA_PATTERNS]
    rng = random.Random(23)
    base = bytearray(rng.choice(b"acgt") for _ in range(400000))
    clean = bytes(base)
    want_clean = oracle_counts(oracle, W.REGEXDNA_PATTERNS, clean)
    t_clean = device_text(clean)
    mc = rj.MultiScan(progs)
    assert mc.set_counts_only(True)
    assert mc.run(t_clean.data_ptr(), len(clean)) == want_clean and mc.how == 3
    for chain in (b"agggtaaagggtaaagggtaaa", b"agggtaaagggtaaagggtaaagggtaaagggtaaa", b"tttaccctttaccctttaccct"):
        t = bytearray(base)
        t[7000:7000 + len(chain)] = chain
        data = bytes(t)
        want = oracle_counts(oracle, W.REGEXDNA_PATTERNS, data)
        c, how, s, bc, bs = both_ways(rj, progs, data)
        assert c == want and s == want, (chain, how, c, s, want)
        assert how == 3 and bc == bs, (chain, how)
    # a pair at EVERY block boundary: some of them lie across two waves' spans, whatever the launch geometry
    t = bytearray(base)
    for edge in range(2048, len(t) - 64, 2048):
        t[edge - 4:edge + 11] = b"agggtaaagggtaaa"
    data = bytes(t)
    want = oracle_counts(oracle, W.REGEXDNA_PATTERNS, data)
    tt = device_text(data)
    assert mc.run(tt.data_ptr(), len(data)) == want, mc.how
    assert mc.how == 1
    bounds_counts = mc.bounds()
    ms = rj.MultiScan(progs)
    assert ms.run(tt.data_ptr(), len(data)) == want and ms.bounds() == bounds_counts
    assert mc.run(t_clean.data_ptr(), len(clean)) == want_clean and mc.how == 3


def test_match_all_count_of_one_pattern_takes_the_kernel(rj, oracle, W):
    """The reference's own entry point: regexdna calls the single-pattern MatchAllCount nine times (sample/regexdna.cc:65,
    src/rejit.cc:84-86,203-208).  rj_match_all(prog, text, n, NULL) -- host text -- and rj_scan_count -- device text --
    must run plane_count for every one of the nine (rj_stats.count_path), with the oracle's counts; a pattern without the
    shape takes the pipeline; there is no span list to copy or replace afterwards."""
    data = W.fasta_stripped_numpy(50000).tobytes()
    pair = b"agggtaaagggtaaa"
    t2 = bytearray(data[:200000])
    for at in (5000, 2048 * 20 + 700, 150000):      # (mid-block: texts this small give every wave ONE block)
        t2[at:at + len(pair)] = pair
    t2[70000:70022] = b"agggtaaagggtaaagggtaaa"        # a chain: walked inside the kernel
    texts = [data, bytes(t2[:60000]), bytes(t2)]
    for k, text in enumerate(texts):
        tt = device_text(text)
        for rx in W.REGEXDNA_PATTERNS:
            want = len(oracle.match_all(rx.encode(), text))
            prog = rj.Program(rx)
            assert prog.count(text) == want, (k, rx)
            hs = prog.host_stats()
            assert hs["n_matches"] == want
            scan = rj.Scan(prog)
            assert scan.count_tensor(tt) == want, (k, rx)
            st = scan.stats()
            assert st["count_path"] == 1, (k, rx, st)
            assert hs["count_path"] == st["count_path"], (k, rx, hs)
            if st["count_path"] == 1 and want:
                assert scan.device_spans_ptr() == 0
                with pytest.raises(rj.RejitError):
                    scan.spans()
            # the ordinary run of the same scan object afterwards: lists again
            assert scan.run_tensor(tt) == want and scan.spans() == oracle.match_all(rx.encode(), text)
    # other literals take the general form of the kernel; unbounded patterns and assertions the pipeline (count_path 0)
    for rx, path in ((b"agggtaaac", 1), (b"regexp", 1), (b"[acgt]+x", 0), (b"^agggtaaa", 0), (b"a.*b", 0), (b"agg", 0)):
        prog = rj.Program(rx)
        text = texts[1] + b"agggtaaacxx regexp acgtx\nagggtaaa"
        assert prog.count(text) == len(oracle.match_all(rx, text)), rx
        assert prog.host_stats()["count_path"] == path, rx
    # small and empty texts
    prog = rj.Program(W.REGEXDNA_PATTERNS[0])
    for text in (b"", b"agggtaaa", b"xxagggtaaaxxtttaccct", b"agggtaaagggtaaa"):
        assert prog.count(text) == len(oracle.match_all(W.REGEXDNA_PATTERNS[0].encode(), text)), text


def test_a_block_full_of_candidates_voids_the_run(rj, oracle, W):
    """`agggtaaa` back to back: a candidate every 8 bytes, 256 per 2-KiB block.  A wave with one block classifies them all
    (matches 8 bytes apart do not overlap: the counts stand); a wave with several blocks overfills its ring between two
    looks at it: the run is void and the span pipeline answers."""
    progs = [rj.Program(rx) for rx in W.REGEXDNA_PATTERNS]
    for reps, void in ((5000, False), (400000, True)):
        data = b"agggtaaa" * reps + b"acgt" * 1000
        want = [reps if i == 0 else 0 for i in range(len(progs))]
        if reps <= 5000:
            assert want == oracle_counts(oracle, W.REGEXDNA_PATTERNS, data)
        tt = device_text(data)
        mc = rj.MultiScan(progs)
        mc.set_counts_only(True)
        assert mc.run(tt.data_ptr(), len(data)) == want, reps
        assert mc.how == (1 if void else 3), (reps, mc.how)


def test_counts_two_in_flight(rj, oracle, W):
    progs = [rj.Program(rx) for rx in W.REGEXDNA_PATTERNS]
    data = W.fasta_stripped_numpy(20000).tobytes()
    want = oracle_counts(oracle, W.REGEXDNA_PATTERNS, data)
    tt = device_text(data)
    import torch
    st = torch.cuda.current_stream().cuda_stream
    ms = [rj.MultiScan(progs) for _ in range(2)]
    for m in ms:
        assert m.set_counts_only(True)
    got = []
    ms[0].start(tt.data_ptr(), len(data), stream=st)
    for k in range(1, 12):
        ms[k % 2].start(tt.data_ptr(), len(data), stream=st)
        got.append(ms[(k - 1) % 2].finish())
        assert ms[(k - 1) % 2].how == 3
    got.append(ms[11 % 2].finish())
    assert all(g == want for g in got), got[:3]


def test_other_set_shapes(rj, oracle):
    """One base; classes with bytes >= 0x80; a set the plan refuses (the switch is then ignored)."""
    rng = random.Random(14)
    rxs = [b"abcdefgh", b"abc[\x80-\xff]efgh", b"xbcdefgh", b"abcdef[^g]h"]
    progs = [rj.Program(rx) for rx in rxs]
    strings = [b"abcdefgh", b"abc\x80efgh", b"abc\xffefgh", b"abcdefxh", b"abcdef\x00h", b"xbcdefgh", b"abcdeggh", b"abcdefgg"]
    data = bytes(planted_text(rng, 120000, bytes(range(256)), strings, 0.002))
    want = oracle_counts(oracle, rxs, data)
    tt = device_text(data)
    mc = rj.MultiScan(progs)
    took = mc.set_counts_only(True)
    c = mc.run(tt.data_ptr(), len(data))
    assert c == want, (took, mc.how, c, want)
    ms = rj.MultiScan(progs)
    assert ms.run(tt.data_ptr(), len(data)) == want
    # a longer pattern in the set: not the 8-mer table's shape -- the general form of the kernel takes it (round 6)
    progs2 = [rj.Program(rx) for rx in (b"agggtaaa|tttaccct", b"agggtaaac")]
    m2 = rj.MultiScan(progs2)
    assert m2.set_counts_only(True)
    data2 = b"xxagggtaaacxxtttaccctxx" * 100
    t2 = device_text(data2)
    assert m2.run(t2.data_ptr(), len(data2)) == oracle_counts(oracle, [b"agggtaaa|tttaccct", b"agggtaaac"], data2) and m2.how == 3
    # an unbounded pattern in the set: the switch is ignored
    progs3 = [rj.Program(rx) for rx in (b"agggtaaa|tttaccct", b"agggtaaa[acgt]+x")]
    m3 = rj.MultiScan(progs3)
    assert not m3.set_counts_only(True)
    data3 = b"xxagggtaaacxxtttaccctxx" * 100
    t3 = device_text(data3)
    assert m3.run(t3.data_ptr(), len(data3)) == oracle_counts(oracle, [b"agggtaaa|tttaccct", b"agggtaaa[acgt]+x"], data3) and m3.how != 3


def _sharded_counts(rj, oracle, patterns, text, world, halo, cut_align=1024):
    """rj_multi_device_counts_via over `world` shards of one text on the one device (a thread, a stream and an rj_multi per shard,
    the all-gather played by a barrier and device-to-device copies, as tests/test_gpu_device_counts.py), every shard's object in
    counts mode; returns (per-rank counts, per-rank `how`)."""
    import ctypes
    import threading
    import torch
    from rejit_amd import api, sharding
    dev = torch.device("cuda:0")
    n = len(text)
    ranges = sharding.partition(n, world, align=cut_align)
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    barrier = threading.Barrier(world)
    sends = [None] * world
    results, hows, errors = [None] * world, [None] * world, []

    def rank_main(rank):
        try:
            own = ranges[rank]
            lo, hi = sharding.visible_range(n, own, halo)
            shard = torch.frombuffer(bytearray(text[lo:hi]), dtype=torch.uint8).to(dev)
            stream = torch.cuda.Stream(dev)
            multi = rj.MultiScan([rj.Program(p) for p in patterns])
            assert multi.set_counts_only(True)

            def allgather(ctx, send, recv, nbytes, st):
                if hip.hipStreamSynchronize(st) != 0:
                    return 1
                sends[rank] = send
                barrier.wait(timeout=120)
                bad = 0
                for r in range(world):
                    bad |= hip.hipMemcpyAsync(recv + r * nbytes, sends[r], nbytes, 3, st)   # hipMemcpyDeviceToDevice
                bad |= hip.hipStreamSynchronize(st)
                barrier.wait(timeout=120)
                return bad

            cb = api.ALLGATHER_FN(allgather)
            for _ in range(2):
                results[rank] = multi.device_counts(shard.data_ptr(), hi - lo, lo, rank, world, allgather=cb, own_begin=own[0] - lo,
                                                    own_end=own[1] - lo, stream=stream.cuda_stream)
                hows[rank] = multi.how
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    return results, hows, ranges


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_counts_in_counts_mode(rj, oracle, W, world):
    """Every shard's object in counts mode: the rows of the carry exchange come from the kernel's first / last match per pattern.
    One cut has a pair of overlapping matches of pattern 0 across it (the right shard must re-select under the carry), one
    cut goes through a match."""
    from rejit_amd import sharding
    rng = random.Random(15 + world)
    n = 300000
    t = spaced_text(rng, n, b"ac", regexdna_strings())   # (background of a / c only: no match outside the planted strings)
    ranges = sharding.partition(n, world, align=1024)
    cut1 = ranges[0][1]
    t[cut1 - 4:cut1 - 4 + 15] = b"agggtaaagggtaaa"          # first match begins left of the cut, the second right of it
    for i in range(cut1 - 60, cut1 - 4):                      # (keep other matches of pattern 0 away from the pair)
        t[i] = ord("c")
    for i in range(cut1 + 11, cut1 + 60):
        t[i] = ord("c")
    if world > 2:
        cut2 = ranges[1][1]
        t[cut2 - 3:cut2 + 5] = b"tttaccct"
    text = bytes(t)
    patterns = W.REGEXDNA_PATTERNS
    want = oracle_counts(oracle, patterns, text)
    results, hows, _ = _sharded_counts(rj, oracle, patterns, text, world, 8)
    for rank in range(world):
        assert results[rank] == want, (rank, results[rank], want)
        assert hows[rank] == 3, (rank, hows)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_counts_of_a_general_set(rj, oracle, W, world):
    """The same exchange with the GENERAL form of the count kernel (round 6): literals of different lengths, so the rows carry
    first / last matches with their own lengths; one cut goes through a long match, one has two overlapping matches of one pattern
    (`abcabcabc` inside `abcabcabcabc...`) across it -- the right shard's first match begins inside the left shard's last."""
    from rejit_amd import sharding
    rng = random.Random(25 + world)
    patterns = ["alternation|strings", "prefix abcd|prefix 1234", "abcabcabc"]
    strings = [b"alternation", b"strings", b"prefix abcd", b"prefix 1234", b"abcabcabc", b"alternatio", b"prefix 12"]
    n = 300000
    t = spaced_text(rng, n, b"xyz 09", strings)
    ranges = sharding.partition(n, world, align=1024)
    cut1 = ranges[0][1]
    for i in range(cut1 - 80, cut1 + 80):
        t[i] = ord("x")
    t[cut1 - 5:cut1 - 5 + 15] = b"abcabcabcabcabc"            # matches of `abcabcabc` at -5 (kept), -2 and +1 (inside it)
    if world > 2:
        cut2 = ranges[1][1]
        for i in range(cut2 - 40, cut2 + 40):
            t[i] = ord("y")
        t[cut2 - 6:cut2 + 5] = b"alternation"
    text = bytes(t)
    want = oracle_counts(oracle, patterns, text)
    results, hows, _ = _sharded_counts(rj, oracle, patterns, text, world, 16)
    for rank in range(world):
        assert results[rank] == want, (rank, results[rank], want)
        assert hows[rank] == 3, (rank, hows)


# ------------------------------------------------------------------------------------------------------------------
# Round 6: MatchAllCount in one kernel for the sets the GENERAL plan takes (plane_count.hip: GeneralShape) -- alternations
# of literals of any length, k-mers of any k, windows with a class position (the reference's fast forward takes any
# alternation of <= 7 literals, src/x64/codegen-x64.cc:1129-1252, src/codegen.cc:327-393).

def general_both_ways(rj, rxs, data, own=None, expect_how=3):
    progs = [rj.Program(rx) for rx in rxs]
    t = device_text(data)
    kw = {} if own is None else {"own_begin": own[0], "own_end": own[1]}
    mc = rj.MultiScan(progs)
    took = mc.set_counts_only(True)
    c = mc.run(t.data_ptr(), len(data), **kw)
    how, bc = mc.how, mc.bounds()
    ms = rj.MultiScan(progs)
    s = ms.run(t.data_ptr(), len(data), **kw)
    bs = ms.bounds()
    if expect_how is not None:
        assert took and how == expect_how, (rxs, took, how)
    return c, how, s, bc, bs


def random_kmers(rng, k, count, alphabet):
    out = set()
    while len(out) < count:
        out.add(bytes(rng.choice(alphabet) for _ in range(k)))
    return sorted(out)


def test_general_counts_literal_sets(rj, oracle, W):
    """bench.py's general_one_pass set, nine random 6-mers and 12-mers (DNA alphabet, DNA text: matches and close pairs
    are common; lower-case letters over random ASCII with plants), mixed lengths -- counts == oracle, first / last matches
    == the span pipeline's, return value 3."""
    rng = random.Random(31)
    ascii_bg = bytes(range(ord("0"), ord("z")))
    cases = []
    gset = [b"alternation|strings", b"prefix abcd|prefix 1234"]
    gstrings = [b"alternation", b"strings", b"prefix abcd", b"prefix 1234", b"alternatio", b"prefix abc4", b"string", b"prefix 123"]
    cases.append((gset, planted_text(rng, 300000, ascii_bg, gstrings, 0.002)))
    for k in (6, 12):
        dna = random_kmers(rng, k, 9, b"acgt")
        text = bytearray(np.random.default_rng(k).choice(np.frombuffer(b"acgt", dtype=np.uint8), 400000).tobytes())
        for s in dna:                                   # (12-mers do not occur by chance)
            for _ in range(40):
                p = rng.randrange(0, len(text) - k)
                text[p:p + k] = s
        cases.append((dna, text))
        words = random_kmers(rng, k, 9, b"abcdefghijklmnopqrstuvwxyz")
        cases.append((words, planted_text(rng, 300000, ascii_bg, words + [w[:-1] for w in words], 0.003)))
    mixed = [b"needle", b"haystacks|haycart", b"pitchfork12", b"barn door"]   # (five bases of >= 6 compared bytes: the plan's filter is selective enough)
    cases.append((mixed, planted_text(rng, 300000, ascii_bg, [b"needle", b"haystacks", b"haystack", b"haycart", b"pitchfork12", b"pitchfork1", b"barn door"], 0.003)))
    for rxs, t in cases:
        data = bytes(t)
        want = oracle_counts(oracle, rxs, data)
        c, how, s, bc, bs = general_both_ways(rj, rxs, data)
        assert s == want, (rxs, s, want)
        assert c == want, (rxs, how, c, want)
        assert bc == bs, (rxs, bc, bs)
        assert sum(want) > 0


def test_general_counts_classes_offsets_and_prefixes(rj, oracle, W):
    """Windows with a class position (tolerance 1), windows behind a class (their own offset inside the match), an
    alternative that is a prefix of another (the longest wins and is what the next match must not begin inside), own ranges."""
    rng = random.Random(32)
    ascii_bg = bytes(range(ord("0"), ord("z")))
    cases = [
        ([b"ab[cx]defgh", b"zzz[0-9]yyyy"], [b"abcdefgh", b"abxdefgh", b"abydefgh", b"zzz5yyyy", b"zzzayyyy", b"zzz0yyy"]),
        ([b"[ab]cdefghij", b"[xy]cdefghiq"], [b"acdefghij", b"bcdefghij", b"ccdefghij", b"xcdefghiq", b"ycdefghij"]),
        ([b"abcd|abcdefgh", b"efgh1234"], [b"abcd", b"abcdefgh", b"abcdefgh1234", b"abcdabcd", b"efgh1234"]),
        ([b"ab(cd|ef)ghij", b"abcdgh"], [b"abcdghij", b"abefghij", b"abcdgh", b"abcdghi"]),
    ]
    for rxs, strings in cases:
        data = bytes(planted_text(rng, 250000, ascii_bg, strings, 0.004))
        want = oracle_counts(oracle, rxs, data)
        c, how, s, bc, bs = general_both_ways(rj, rxs, data, expect_how=None)
        assert s == want and c == want, (rxs, how, c, s, want)
        assert bc == bs, (rxs, how, bc, bs)
        assert how == 3, (rxs, "the general plan takes this set: the count must be the kernel's")
        for own in [(0, 100000), (100000, 250001), (2040, 2060), (123457, 123458)]:
            want_own = oracle_counts(oracle, rxs, data, own)
            c, how, s, bc, bs = general_both_ways(rj, rxs, data, own, expect_how=None)
            assert c == want_own and s == want_own, (rxs, own, how, c, s, want_own)
            assert bc == bs, (rxs, own)


def test_general_counts_selection_along_chains(rj, oracle, W):
    """Texts on which the count is far from the number of matching positions: `abab|baba` over `abababab...` (every position
    matches, every second one overlaps the one before), runs of a 6-mer's period, and a dense mix -- the selection is walked
    inside the kernel (a pair or a chain inside a span never voids the run; across two waves' spans it may)."""
    rng = random.Random(33)
    rxs = [b"abab|baba", b"ababab"]
    data = b"ab" * 3000 + b"xx" + b"ba" * 1000 + b"x" * 2000 + b"abab" + b"y" * 100 + b"ababab" + b"z" * 3000
    want = oracle_counts(oracle, rxs, data)
    c, how, s, bc, bs = general_both_ways(rj, rxs, data, expect_how=None)
    assert s == want, (s, want)
    assert c == want, (how, c, want)
    assert bc == bs
    # bursts of `abab...` of every length up to 40 with fillers between them (a wave's ring takes 256 candidates per two
    # blocks): pairs and chains inside a span are the kernel's own business
    small = b"".join(b"ab" * (k % 20 + 1) + b"q" * 300 + b"ba" * (k % 7 + 2) + b"x" * 250 for k in range(40))
    small = small[:2000]        # (texts below 2 MiB give every wave ONE block: stay inside the first)
    want = oracle_counts(oracle, rxs, small)
    c, how, s, bc, bs = general_both_ways(rj, rxs, small, expect_how=None)
    assert c == want and s == want and how == 3 and bc == bs, (how, c, s, want)
    # a dense mix over two letters: six patterns of 5..9 bytes
    pats = [b"aabab", b"babbab", b"abbabba", b"bbbbbbbb", b"aaaaaaaab", b"abaab|baaba"]
    data = bytes(rng.choice(b"ab") for _ in range(150000))
    want = oracle_counts(oracle, pats, data)
    c, how, s, bc, bs = general_both_ways(rj, pats, data, expect_how=None)
    assert c == want and s == want, (how, c, s, want)
    assert bc == bs


def test_general_counts_single_pattern_and_host_entry(rj, oracle, W):
    """MatchAllCount of ONE pattern of the general shape: `alternation|strings` through rj_scan_count and through
    rj_match_all(..., NULL)."""
    rng = random.Random(34)
    ascii_bg = bytes(range(ord("0"), ord("z")))
    for rx, strings in [(b"alternation|strings", [b"alternation", b"strings", b"string"]), (b"prefix abcd|prefix 1234", [b"prefix abcd", b"prefix 1234", b"prefix 12"]),
                        (b"acgtacgtacgt", [b"acgtacgtacgt", b"acgtacgtacg"])]:
        data = bytes(planted_text(rng, 200000, ascii_bg, strings, 0.003))
        want = len(oracle.match_all(rx, data))
        prog = rj.Program(rx)
        assert prog.count(data) == want
        assert prog.host_stats()["count_path"] == 1, rx
        sc = rj.Scan(prog)
        assert sc.count_tensor(device_text(data)) == want and sc.stats()["count_path"] == 1
