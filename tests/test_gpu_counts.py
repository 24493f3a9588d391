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


def test_chains_and_cut_pairs_void_THAT_run_only(rj, oracle, W):
    """Three matches in a row, each inside the one before (`agggtaaagggtaaagggtaaa`: the reference keeps the first and the
    third), and a pair that lies across two waves' spans: the kernel flags the run and the span pipeline answers (return
    value 1, the oracle's counts) -- and the NEXT run of the same object, on a clean text, is back on the counts path
    (round 5 left the object on the span pipeline for good)."""
    progs = [rj.Program(rx) for rx in W.REGEXDNA_PATTERNS]
    rng = random.Random(23)
    base = bytearray(rng.choice(b"acgt") for _ in range(400000))
    clean = bytes(base)
    want_clean = oracle_counts(oracle, W.REGEXDNA_PATTERNS, clean)
    t_clean = device_text(clean)
    mc = rj.MultiScan(progs)
    assert mc.set_counts_only(True)
    assert mc.run(t_clean.data_ptr(), len(clean)) == want_clean and mc.how == 3
    t = bytearray(base)
    t[7000:7022] = b"agggtaaagggtaaagggtaaa"
    data = bytes(t)
    want = oracle_counts(oracle, W.REGEXDNA_PATTERNS, data)
    tt = device_text(data)
    assert mc.run(tt.data_ptr(), len(data)) == want, mc.how
    assert mc.how == 1, "a chain of three must send the run to the span pipeline"
    assert mc.run(t_clean.data_ptr(), len(clean)) == want_clean and mc.how == 3, "the fallback is per run, not per object"
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
    t2[70000:70022] = b"agggtaaagggtaaagggtaaa"        # a chain: this text takes the pipeline, with the same answer
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
            chain_hits = k == 2 and rx == W.REGEXDNA_PATTERNS[0]
            assert st["count_path"] == (0 if chain_hits else 1), (k, rx, st)
            assert hs["count_path"] == st["count_path"], (k, rx, hs)
            if st["count_path"] == 1 and want:
                assert scan.device_spans_ptr() == 0
                with pytest.raises(rj.RejitError):
                    scan.spans()
            # the ordinary run of the same scan object afterwards: lists again
            assert scan.run_tensor(tt) == want and scan.spans() == oracle.match_all(rx.encode(), text)
    # not the shape: the pipeline answers, count_path stays 0
    for rx in (b"agggtaaac", b"[acgt]+x", b"regexp"):
        prog = rj.Program(rx)
        text = texts[1] + b"agggtaaacxx regexp acgtx"
        assert prog.count(text) == len(oracle.match_all(rx, text))
        assert prog.host_stats()["count_path"] == 0
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
    # a longer pattern in the set: not the shape
    progs2 = [rj.Program(rx) for rx in (b"agggtaaa|tttaccct", b"agggtaaac")]
    m2 = rj.MultiScan(progs2)
    assert not m2.set_counts_only(True)
    data2 = b"xxagggtaaacxxtttaccctxx" * 100
    t2 = device_text(data2)
    assert m2.run(t2.data_ptr(), len(data2)) == oracle_counts(oracle, [b"agggtaaa|tttaccct", b"agggtaaac"], data2)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_counts_in_counts_mode(rj, oracle, W, world):
    """rj_multi_device_counts_via over shards of one text on the one device (a thread, a stream and an rj_multi per shard,
    the all-gather played by a barrier and device-to-device copies, as tests/test_gpu_device_counts.py), every shard's
    object in counts mode: the rows of the carry exchange come from the kernel's first / last match per pattern.  One
    cut has a pair of overlapping matches of pattern 0 across it (the right shard must re-select under the carry), one
    cut goes through a match."""
    import ctypes
    import threading
    import torch
    from rejit_amd import api, sharding
    dev = torch.device("cuda:0")
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
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    barrier = threading.Barrier(world)
    sends = [None] * world
    results, hows, errors = [None] * world, [None] * world, []

    def rank_main(rank):
        try:
            own = ranges[rank]
            lo, hi = sharding.visible_range(n, own, 8)
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
    for rank in range(world):
        assert results[rank] == want, (rank, results[rank], want)
        assert hows[rank] == 3, (rank, hows)
