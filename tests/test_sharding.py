"""CPU tests of the multi-rank path: world_size-2 (and 3) gloo process groups run the
sharding protocol of rejit_amd/sharding.py with the oracle as the local matcher, and the
gathered result must equal the single-rank result -- including patterns whose matches
overlap the cut, so that the carry exchange is exercised."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from checkers import Oracle
from rejit_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_scan_factory(ends):
    def scan(lo, hi, cur, prev_end, have):
        out = []
        for s in range(lo, min(hi, len(ends))):
            e = ends[s]
            if e < 0 or s < cur:
                continue
            cur = e if e > s else s + 1
            if not (e == s and have and prev_end == s):
                out.append((s, e))
            have, prev_end = True, e
        return out
    return scan


CASES = [(b"abc", 1), (b"aa", 1), (b"(ab|ba)+", 1), (b"x*", 1), (b"a[bc]*a", 1), (b"^|b$", 1), (b"aaa", 1), (b"a{5}", 1)]


def _worker(rank, world, port, text, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = Oracle()
    results = {}
    for rx, align in CASES:
        ends = oracle.longest_all(rx, text)
        ranges = sharding.partition(len(text), world, align=align)
        total, gathered, local = sharding.sharded_match_all(_local_scan_factory(ends), ranges, rank, world, dist)
        results[rx] = (total, gathered)
        # the tensor form (what bench.py's multi-GPU workloads use): same protocol, no tuple lists
        import torch
        scan_l = _local_scan_factory(ends)
        tscan = lambda lo, hi, cur, pe, have: torch.tensor(scan_l(lo, hi, cur, pe, have), dtype=torch.int64).reshape(-1, 2)
        t_total, t_gathered, _ = sharding.sharded_match_all_tensor(tscan, ranges, rank, world, dist, torch.device("cpu"))
        assert t_total == total
        if rank == 0:
            assert [tuple(x) for x in t_gathered.tolist()] == gathered
    # several patterns' COUNTS with the selection carried over the cuts (bench.py's regexdna step)
    pats = [rx for rx, _ in CASES]
    all_ends = [oracle.longest_all(rx, text) for rx in pats]
    ranges = sharding.partition(len(text), world, align=1)
    own = ranges[rank]

    def bounds_of(sp):
        return (sp[0][0], sp[0][1], sp[-1][0], sp[-1][1]) if sp else None

    def run_local():
        sps = [_local_scan_factory(e)(own[0], own[1], 0, 0, False) for e in all_ends]
        return [len(sp) for sp in sps], [bounds_of(sp) for sp in sps]

    def rerun_one(i, cur, prev_end):
        sp = _local_scan_factory(all_ends[i])(own[0], own[1], cur, prev_end, True)
        return len(sp), bounds_of(sp)

    import torch
    results["multi_counts"] = sharding.multi_pattern_counts(run_local, rerun_one, len(pats), rank, world, dist, torch.device("cpu"))
    if rank == 0:
        q.put(results)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "random"), (3, "random"), (3, "run"), (4, "run")])
def test_sharded_equals_single(world, kind):
    rng = random.Random(world)
    if kind == "random":
        text = bytes(rng.choice(b"aabbc\nx") for _ in range(997))
    else:
        # self-overlapping occurrences over every cut: a rank re-runs with its left neighbour's EMPTY-carry last
        # match while that neighbour re-runs too and its last match moves earlier -- the rank must select again
        # under the smaller carry (the round-2 protocol compared its CURRENT first match and skipped starts)
        text = b"a" * 1003 + b"b" + b"a" * 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, text, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oracle = Oracle()
    for rx, _ in CASES:
        want = oracle.match_all_spec(rx, text)
        total, gathered = results[rx]
        assert total == len(want) and gathered == want, rx
    assert results["multi_counts"] == [len(oracle.match_all_spec(rx, text)) for rx, _ in CASES]


def test_partition_and_visible_range():
    r = sharding.partition(10_000_000, 8)
    assert r[0][0] == 0 and r[-1][1] == 10_000_001
    assert all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
    assert all(lo % 1024 == 0 for lo, _ in r)
    assert sharding.visible_range(1000, (100, 200), 8) == (36, 208)
    assert sharding.visible_range(1000, (100, 200), None) == (36, 1000)
    assert sharding.visible_range(1000, (100, 200), 8, whole_text=True) == (0, 1000)
    assert sharding.needs_rerun((5, 9), (6, 6, True)) and not sharding.needs_rerun((6, 9), (6, 6, True))
    assert sharding.needs_rerun((6, 6), (6, 6, True)) and not sharding.needs_rerun(None, (6, 6, True))


# ----------------------------------------------------------------------------- file sharding (jrep, C5)
def test_partition_files_balances_and_is_deterministic():
    rng = random.Random(2)
    sizes = [rng.choice([0, 10, 500, 4000, 100000, 2000000]) for _ in range(500)]
    for world in (1, 2, 3, 8):
        parts = sharding.partition_files(sizes, world)
        assert sorted(i for p in parts for i in p) == list(range(len(sizes)))       # every file exactly once
        assert all(p == sorted(p) for p in parts)                                      # original order kept
        loads = [sum(sizes[i] + 1 for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(sizes) + 1                               # greedy bound
        assert parts == sharding.partition_files(list(sizes), world)


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = (b"rank%d:" % rank) + bytes(range(rank * 7 % 256, 256)) * rank    # different lengths, rank 0 short
    whole = sharding.gather_bytes(blob, rank, world, dist)
    empty = sharding.gather_bytes(b"", rank, world, dist)
    if rank == 0:
        q.put((whole, empty))
    else:
        assert whole is None and empty is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_bytes_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    whole, empty = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = b"".join((b"rank%d:" % r) + bytes(range(r * 7 % 256, 256)) * r for r in range(world))
    assert whole == want and empty == b""
