"""GPU parity at BASELINE's FULL sizes (-m gpu): the engine's spans against the REAL reference's answers over the same
inputs, stored as counts + span digests in tests/golden/fullsize_vectors.json (generated once in the build container by
tests/golden/make_fullsize.py from oracle/_ref, the reference compiled in place).
  C3  the nine regexdna patterns over the stripped 50M-line FASTA (500 MB): one-pass run (plane scan) and single runs;
      MatchAllCount in one kernel (what bench.py times), own ranges cut mid-block, and a 4.3 GB text (32-bit offsets)
  C2  literal `regexp` over 5 GB of random ASCII with 1000 planted occurrences
  C4  the complex benchmark regex over rank 0's 6.25 GB shard of the 8-GPU job (own range + 58-byte halo), and (round 4)
      ALL eight shards one after the other with the selection carried over the seven cuts
  C5  bench.py's 100 000-file / 10 GB tree: per-file match and line-start counts against the reference's jrep logic"""
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def env():
    import torch
    import rejit_amd
    from rejit_amd import workloads as W
    rejit_amd.build()
    with open(os.path.join(HERE, "golden", "fullsize_vectors.json")) as fh:
        doc = json.load(fh)
    dev = torch.device("cuda:0")
    return rejit_amd, W, torch, dev, doc


def digest_of(W, scan, dev):
    return W.span_digest_torch(scan.spans_tensor(dev))


def test_c3_regexdna_50m_lines(env):
    rj, W, torch, dev, doc = env
    c3 = doc["c3"]
    st = torch.cuda.current_stream(dev).cuda_stream
    text = W.fasta_stripped_torch(c3["fasta_n"], dev)
    n = int(text.numel())
    assert n == c3["bytes"]
    progs = [rj.Program(p["regex"]) for p in c3["patterns"]]
    multi = rj.MultiScan(progs)
    counts = multi.run(text.data_ptr(), n, stream=st)
    assert multi.how == 1          # the one-pass run
    assert counts == [p["digest"]["count"] for p in c3["patterns"]]
    for i, p in enumerate(c3["patterns"]):
        assert digest_of(W, multi.scan(i), dev) == p["digest"], p["regex"]
    # every pattern on its own (window scan + verify)
    for i in (0, 3, 8):
        sc = rj.Scan(progs[i])
        assert sc.run(text.data_ptr(), n, stream=st) == c3["patterns"][i]["digest"]["count"]
        assert digest_of(W, sc, dev) == c3["patterns"][i]["digest"]


def test_c3_counts_only_50m_lines(env):
    """The kernel bench.py times (plane_count: MatchAllCount of the nine patterns in one launch, rj_multi_set_counts_only)
    at BASELINE's size, pinned by the suite and not by a benchmark assert: the real reference's nine counts, the span
    pipeline's first / last match per pattern, own ranges that cut the 500 MB text in the middle of a 2-KiB block (the
    sharded shape), and the same through the reference's own entry point -- MatchAllCount pattern by pattern."""
    rj, W, torch, dev, doc = env
    c3 = doc["c3"]
    st = torch.cuda.current_stream(dev).cuda_stream
    text = W.fasta_stripped_torch(c3["fasta_n"], dev)
    n = int(text.numel())
    want = [p["digest"]["count"] for p in c3["patterns"]]
    progs = [rj.Program(p["regex"]) for p in c3["patterns"]]
    counting, listing = rj.MultiScan(progs), rj.MultiScan(progs)
    assert counting.set_counts_only(True)
    assert counting.run(text.data_ptr(), n, stream=st) == want
    assert counting.how == 3, "the nine patterns must take the one-kernel count at 500 MB"
    assert listing.run(text.data_ptr(), n, stream=st) == want and listing.how == 1
    assert counting.bounds() == listing.bounds()
    # two in flight, as the headline loop runs it
    other = rj.MultiScan(progs)
    assert other.set_counts_only(True)
    counting.start(text.data_ptr(), n, stream=st)
    other.start(text.data_ptr(), n, stream=st)
    assert counting.finish() == want and other.finish() == want and counting.how == 3 and other.how == 3
    # shards: cuts in the middle of a block, and one a few bytes before a block boundary
    for cut in (250000123, 2048 * 100000 - 3, n - 1000):
        total = [0] * len(progs)
        for own in ((0, cut), (cut, n + 1)):
            c = counting.run(text.data_ptr(), n, stream=st, own_begin=own[0], own_end=own[1])
            assert counting.how == 3, (cut, own)
            s = listing.run(text.data_ptr(), n, stream=st, own_begin=own[0], own_end=own[1])
            assert c == s, (cut, own, c, s)
            assert counting.bounds() == listing.bounds(), (cut, own)
            total = [a + b for a, b in zip(total, c)]
        # (no regexdna match overlaps another of its pattern in this text: the two ranges' counts add up)
        assert total == want, (cut, total, want)
    # the reference's own call: one MatchAllCount per pattern (sample/regexdna.cc:65) -- device text
    for i, p in enumerate(progs):
        sc = rj.Scan(p)
        assert sc.count(text.data_ptr(), n, stream=st) == want[i], c3["patterns"][i]["regex"]
        assert sc.stats()["count_path"] == 1


def test_counts_only_beyond_4gib(env):
    """32-bit arithmetic in the count kernel (block indices, the candidates' offsets inside a span, the workgroup rows)
    on a text of 4.3 GB: the FASTA generator's random sections are periodic (the LCG's period is 139 968 characters), so
    the 43M-line text is more of the same; counts and first / last matches must equal the span pipeline's, a match planted
    in the last 8 bytes (offset > 2^32) must be the last one, and a range that begins beyond 4 GiB must count like the
    span pipeline does."""
    rj, W, torch, dev, doc = env
    st = torch.cuda.current_stream(dev).cuda_stream
    fasta_n = 430000000
    text = W.fasta_stripped_torch(fasta_n, dev)
    n = int(text.numel())
    assert n == 4300000000 and n > (1 << 32)
    W.plant(text, [n - 8], b"agggtaaa")
    W.plant(text, [(1 << 32) - 4], b"tttaccct")     # across the 4 GiB line
    progs = [rj.Program(rx) for rx in W.REGEXDNA_PATTERNS]
    counting, listing = rj.MultiScan(progs), rj.MultiScan(progs)
    assert counting.set_counts_only(True)
    c = counting.run(text.data_ptr(), n, stream=st)
    assert counting.how == 3
    s = listing.run(text.data_ptr(), n, stream=st)
    assert listing.how == 1
    assert c == s, (c, s)
    bc, bs = counting.bounds(), listing.bounds()
    assert bc == bs, (bc, bs)
    assert bc[0][2:] == (n - 8, n)
    # the periodic structure: sequence THREE (the last 5 n characters) repeats every 139 968 characters, so a stretch of
    # whole periods holds the same matches wherever it begins: 30 periods early in the sequence and 30 periods that begin
    # beyond 4 GiB
    period = 139968
    a0 = 5 * fasta_n + 7 * period
    a1 = a0 + (((1 << 32) - a0 + period - 1) // period) * period
    assert a1 >= (1 << 32) and a1 + 30 * period < n - 8
    stretch = [counting.run(text.data_ptr(), n, stream=st, own_begin=a, own_end=a + 30 * period) for a in (a0, a1)]
    assert counting.how == 3
    assert stretch[0] == stretch[1], stretch
    assert sum(stretch[0]) > 0
    # a range beyond 4 GiB
    own = ((1 << 32) + 12345, n + 1)
    assert counting.run(text.data_ptr(), n, stream=st, own_begin=own[0], own_end=own[1]) == listing.run(text.data_ptr(), n, stream=st, own_begin=own[0], own_end=own[1])
    assert counting.how == 3 and counting.bounds() == listing.bounds()
    sc = rj.Scan(progs[0])
    assert sc.count(text.data_ptr(), n, stream=st) == c[0] and sc.stats()["count_path"] == 1


def test_run_kernels_beyond_4gib(env):
    """The run kernels (run_scan.hip: `[acgt]+`, one long-lived thread in one loop position) on the 4.3 GB FASTA text: its last
    2.15 GB are ONE run of acgt, sequence TWO (1.29 GB) is runs of acgt between IUB codes, sequence ONE is upper case.  The runs are
    computed independently with torch, chunk by chunk (count, sum of begins, sum of lengths, the longest), and a range that begins
    beyond 4 GiB must give the tail of the one long run's ... nothing: the run began before the range, so a fresh start at the range's
    begin reports (begin, n) -- the independent-range semantics of include/rejit_hip.h."""
    rj, W, torch, dev, doc = env
    fasta_n = 430000000
    text = W.fasta_stripped_torch(fasta_n, dev)
    n = int(text.numel())
    assert n > (1 << 32)
    lut = torch.zeros(256, dtype=torch.uint8, device=dev)
    lut[torch.tensor(list(b"acgt"), device=dev)] = 1
    count, sum_begin, sum_len, longest, prev_inside, open_begin = 0, 0, 0, 0, False, 0
    step = 1 << 28
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        inside = lut[text[lo:hi].long()].bool()
        prev = torch.cat([torch.tensor([prev_inside], device=dev), inside[:-1]])
        begins = torch.nonzero(inside & ~prev).flatten() + lo
        nxt_zero = torch.nonzero(~inside & prev).flatten() + lo        # ends: the first byte outside a run
        count += int(begins.numel())
        sum_begin += int(begins.sum().item())
        # lengths: pair the ends with the begins in order (a run open at the chunk's begin closes with the first end)
        b_list = ([open_begin] if prev_inside else []) + begins.tolist() if begins.numel() < 2_000_000 else None
        if b_list is None:
            # (sequence TWO: millions of short runs per chunk -- pair them on the device)
            bb = torch.cat([torch.tensor([open_begin], device=dev, dtype=begins.dtype), begins]) if prev_inside else begins
            k = int(nxt_zero.numel())
            lens = nxt_zero - bb[:k]
            sum_len += int(lens.sum().item())
            longest = max(longest, int(lens.max().item()) if k else 0)
            prev_inside = bool(inside[-1].item())
            open_begin = int(bb[k].item()) if prev_inside else 0
        else:
            e_list = nxt_zero.tolist()
            for b, e in zip(b_list, e_list):
                sum_len += e - b
                longest = max(longest, e - b)
            prev_inside = bool(inside[-1].item())
            open_begin = b_list[len(e_list)] if prev_inside else 0
        del inside, prev, begins, nxt_zero
    if prev_inside:
        sum_len += n - open_begin
        longest = max(longest, n - open_begin)
    sc = rj.Scan(rj.Program(b"[acgt]+"))
    cnt = sc.run_tensor(text)
    st = sc.stats()
    assert st["run_path"] == 1 and st["linear_path"] == 0, st
    sp = sc.spans_tensor(dev)
    assert cnt == count == int(sp.shape[0])
    assert int(sp[:, 0].sum().item()) == sum_begin
    lens = sp[:, 1] - sp[:, 0]
    assert int(lens.sum().item()) == sum_len and int(lens.max().item()) == longest and longest > 2_000_000_000
    assert int(sp[-1, 1].item()) == n                                   # the last run ends with the text
    # a range that begins beyond 4 GiB, inside the long run
    ob = (1 << 32) + 4097
    k = sc.run_tensor(text, own_begin=ob, own_end=n + 1)
    assert k == 1 and sc.spans() == [(ob, n)]


def test_c2_literal_5gb(env):
    rj, W, torch, dev, doc = env
    c2 = doc["c2"]
    n, seed = c2["bytes"], c2["seed"]
    t = W.random_ascii_torch(n, seed, dev)
    offs = W.plant_offsets(n, 6, 1000, seed=seed, boundaries=[16, 1024, 1 << 20, 1 << 30, n // 2])
    W.plant(t, offs, b"regexp")
    assert len(offs) == c2["planted"]
    sc = rj.Scan(rj.Program(c2["regex"]))
    assert sc.run(t.data_ptr(), n, stream=torch.cuda.current_stream(dev).cuda_stream) == c2["digest"]["count"]
    assert digest_of(W, sc, dev) == c2["digest"]


def test_c4_complex_shard_of_8(env):
    rj, W, torch, dev, doc = env
    from rejit_amd import sharding
    c4 = doc["c4"]
    world, per = c4["world"], c4["bytes_per_gpu"]
    n_total = per * world
    ranges = sharding.partition(n_total, world)
    own = ranges[0]
    vis_lo, vis_hi = sharding.visible_range(n_total, own, 58)
    assert (vis_lo, vis_hi, own[1]) == (0, c4["visible_bytes"], c4["own_end"])
    t = W.random_ascii_torch(vis_hi, 0xC0FFEE, dev)
    cuts = [r[0] for r in ranges[1:]]
    rng = random.Random(7)
    needles = [(o, W.complex_regex_sample(rng)) for o in W.plant_offsets(n_total, 64, 200 * world, seed=7, boundaries=cuts)]
    for o, s in needles:
        lo, hi = max(o, vis_lo), min(o + len(s), vis_hi)
        if lo < hi:
            W.plant(t, [lo], s[lo - o:hi - o])
    sc = rj.Scan(rj.Program(c4["regex"]))
    k = sc.run(t.data_ptr(), vis_hi, own_begin=0, own_end=own[1], stream=torch.cuda.current_stream(dev).cuda_stream)
    assert k == c4["digest"]["count"]
    assert digest_of(W, sc, dev) == c4["digest"]


def test_c4_all_eight_shards_with_the_carry(env):
    """BASELINE configs[3] at its size, the WHOLE job on one GPU, shard after shard: every one of the eight 6.25 GB shards
    (own range + 58-byte halo, 64 bytes of left context) with the selection carried over the seven cuts exactly as the
    multi-GPU run carries it (the last match before the cut: carry_cur / carry_prev_end), each shard's spans -- global
    offsets -- against the real reference's answer for that shard, and the total against the sum."""
    rj, W, torch, dev, doc = env
    from rejit_amd import sharding
    if "c4all" not in doc:
        pytest.skip("tests/golden/fullsize_vectors.json has no c4all section (make_fullsize.py c4all)")
    c4 = doc["c4all"]
    world, per = c4["world"], c4["bytes_per_gpu"]
    n_total = per * world
    ranges = sharding.partition(n_total, world)
    cuts = [r[0] for r in ranges[1:]]
    rng = random.Random(7)
    needles = [(o, W.complex_regex_sample(rng)) for o in W.plant_offsets(n_total, 64, 200 * world, seed=7, boundaries=cuts)]
    sc = rj.Scan(rj.Program(c4["regex"]))
    st = torch.cuda.current_stream(dev).cuda_stream
    carry = (0, 0, False)   # (cur, prev_end, have) in GLOBAL offsets
    total = 0
    for shard in c4["shards"]:
        r = shard["rank"]
        own = ranges[r]
        vis_lo, vis_hi = sharding.visible_range(n_total, own, 58)
        vis_lo &= ~15
        assert [own[0], min(own[1], n_total + 1)] == shard["own"]
        t = W.random_ascii_torch(vis_hi - vis_lo, 0xC0FFEE, dev, start=vis_lo)
        for o, s in needles:
            lo, hi = max(o, vis_lo), min(o + len(s), vis_hi)
            if lo < hi:
                W.plant(t, [lo - vis_lo], s[lo - o:hi - o])
        have = carry[2] and carry[1] >= vis_lo
        k = sc.run(t.data_ptr(), vis_hi - vis_lo, own_begin=own[0] - vis_lo, own_end=min(own[1], n_total + 1) - vis_lo,
                   carry_cur=max(carry[0] - vis_lo, 0) if have else 0, carry_prev_end=carry[1] - vis_lo if have else 0, have_prev=have, stream=st)
        sp = sc.spans_tensor(dev) + vis_lo
        assert k == shard["digest"]["count"], (r, k, shard["digest"]["count"])
        assert W.span_digest_torch(sp) == shard["digest"], r
        if k:
            b, e = int(sp[-1, 0]), int(sp[-1, 1])
            carry = (e if e > b else b + 1, e, True)
        total += k
        del t, sp
        torch.cuda.empty_cache()
    assert total == c4["total_count"]


def test_c5_jrep_tree_at_baseline_size(env):
    """BASELINE configs[4] at its size: bench.py's 100 000-file / 10 GB tree through rj_match_all_batch in 256 MiB batches
    + the `^` line tables of the files with matches -- files with matches, matches and line starts PER FILE against the real
    reference's jrep logic over the same files (sample/jrep.cc:288-294), as totals and a sha256 of the rows."""
    rj, W, torch, dev, doc = env
    if "c5" not in doc:
        pytest.skip("tests/golden/fullsize_vectors.json has no c5 section (make_fullsize.py c5)")
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    c5 = doc["c5"]
    files = bench.jrep_tree(c5["files"], c5["bytes_asked"])
    assert sum(len(f) for f in files) == c5["bytes"]
    prog, sol = rj.Program(c5["regex"].encode()), rj.Program(c5["line_regex"].encode())
    rows, at, n_files = [], 0, len(files)
    while at < n_files:
        b, size = at, 0
        while at < n_files and (at == b or size + len(files[at]) <= (256 << 20)):
            size += len(files[at])
            at += 1
        res = prog.match_all_batch_counts(files[b:at])
        idx = [b + i for i, k in enumerate(res) if k]
        if idx:
            lc = sol.match_all_batch_counts([files[i] for i in idx])
            rows.extend((i, res[i - b], l) for i, l in zip(idx, lc))
    assert (len(rows), sum(r[1] for r in rows), sum(r[2] for r in rows)) == (c5["files_with_matches"], c5["matches"], c5["line_starts"])
    assert bench.jrep_digest(rows) == c5["sha256"]
