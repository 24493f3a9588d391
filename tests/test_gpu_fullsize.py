"""GPU parity at BASELINE's FULL sizes (-m gpu): the engine's spans against the REAL reference's answers over the same
inputs, stored as counts + span digests in tests/golden/fullsize_vectors.json (generated once in the build container by
tests/golden/make_fullsize.py from oracle/_ref, the reference compiled in place).
  C3  the nine regexdna patterns over the stripped 50M-line FASTA (500 MB): one-pass run (plane scan) and single runs
  C2  literal `regexp` over 5 GB of random ASCII with 1000 planted occurrences
  C4  the complex benchmark regex over rank 0's 6.25 GB shard of the 8-GPU job (own range + 58-byte halo)"""
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
