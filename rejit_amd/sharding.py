"""Sharding one text across ranks (one process per GPU, torch.distributed over RCCL/xGMI).

The reference has no text-chunk parallelism at all (one sequential pass per MatchAll,
src/x64/codegen-x64.cc:535-581); its samples only fan whole patterns / whole files over
threads (sample/regexdna-multithread.cc:65-78, sample/jrep.cc:408-493).  Here one buffer is
cut into contiguous byte ranges:

  * rank r OWNS the match begins in [lo_r, hi_r); it sees the text [lo_r - left, hi_r + right)
    where `right` = max match length - 1 (the automaton may read past hi_r) and `left` >= 1
    (the start-of-line context needs one byte before lo_r);
  * scanning needs no collective.  Only the left-most-longest, non-overlapping SELECTION has
    a dependency across the cut: a match selected in rank r-1 that ends after lo_r suppresses
    begins in rank r.  Ranks therefore first select with an empty carry, exchange their
    carry-out (3 integers, all_gather), and a rank re-runs its (cheap) selection only if the
    true carry-in reaches into its first match -- rarely, and never for the benchmark
    patterns;
  * the exchange step proper is tiny: all_reduce(sum) of match counts (8 B per pattern) and a
    gather of (begin,end) pairs to rank 0 (16 B per match).  With xGMI being point-to-point
    a direct gather is used; there is nothing to bucket.

`local_scan` is injected so the protocol can be exercised on CPU with gloo (tests use the
oracle as the local matcher; bench.py and the product use rejit_amd.Scan on the GPU).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

Span = Tuple[int, int]
# local_scan(own_lo, own_hi, carry_cur, carry_prev_end, have_prev) -> ordered GLOBAL spans whose
# begin lies in [own_lo, own_hi)
LocalScan = Callable[[int, int, int, int, bool], List[Span]]


def partition(n: int, world: int, align: int = 1024) -> List[Tuple[int, int]]:
    """Contiguous ranges covering match begins 0..n (inclusive: the empty match at the end
    of the text belongs to the last rank).  Cuts are multiples of `align`."""
    cuts = [0]
    for r in range(1, world):
        c = (n * r // world) // align * align
        cuts.append(max(c, cuts[-1]))
    cuts.append(n + 1)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def visible_range(n: int, own: Tuple[int, int], max_len: Optional[int], left: int = 64, whole_text: bool = False) -> Tuple[int, int]:
    """Bytes a rank must hold to decide every match that begins in `own`.  max_len None =
    unbounded pattern: the rank needs the text up to its end.  whole_text (pass
    `Program.info()["ring_artefact_risk"]`): a pattern whose answer can depend on the reference's ring
    artefact -- the engine then gives a range the whole segments between synchronisation points of the
    reference's loop, which it looks for in the buffer it is given, so the buffer must be the whole text."""
    lo, hi = own
    if whole_text:
        return 0, n
    vis_lo = max(0, lo - left)
    vis_hi = n if max_len is None else min(n, hi + max_len)
    return vis_lo, vis_hi


def carry_out(spans: Sequence[Span], carry_in: Tuple[int, int, bool]) -> Tuple[int, int, bool]:
    """Selection state after a rank's matches: (smallest allowed next begin, end of the
    previous match, whether there is a previous match)."""
    if not spans:
        return carry_in
    b, e = spans[-1]
    return (e if e > b else b + 1, e, True)


def needs_rerun(first: Optional[Span], carry_in: Tuple[int, int, bool]) -> bool:
    """Would the first match selected with an empty carry change under the true carry-in?"""
    cur, prev_end, have = carry_in
    if first is None:
        return False
    b, e = first
    if b < cur:
        return True
    return bool(have and b == e and prev_end == b)  # zero-length rule, src/codegen.cc:65-73


def must_rerun(first_empty: Optional[Span], used: Tuple[int, int, bool], carry_in: Tuple[int, int, bool]) -> bool:
    """Does a shard whose current result was selected under the carry `used` have to select again under
    `carry_in`?  `first_empty` is its first match under the EMPTY carry.  A shard that has not re-run yet
    (used == empty) re-runs iff the true carry reaches into that first match.  One that HAS re-run holds a result
    that depends on the carry it used; its current first match says nothing about what a different -- e.g.
    smaller, because the left neighbour re-ran in the same round and its last match moved -- carry would select
    (pattern `aaa` over a run of a: the matches skipped under the stale carry were missing), so it re-runs whenever
    the carry changed."""
    if carry_in == used:
        return False
    if used != (0, 0, False):
        return True
    return needs_rerun(first_empty, carry_in)


def sharded_match_all(local_scan: LocalScan, ranges: Sequence[Tuple[int, int]], rank: int, world: int,
                      dist=None, gather_to_root: bool = True):
    """Runs the protocol above.  Returns (total_count, spans_on_root_or_None, local_spans).
    `dist` is torch.distributed (initialised) or None for world == 1."""
    own = ranges[rank]
    empty = (0, 0, False)
    spans = local_scan(own[0], own[1], *empty)
    used = empty
    first_empty = spans[0] if spans else None   # the first match under the EMPTY carry (see must_rerun)
    if world > 1:
        import torch

        dev = _device_for(dist)
        for _ in range(world + 1):
            co = carry_out(spans, used)
            mine = torch.tensor([co[0], co[1], int(co[2]), spans[0][0] if spans else -1,
                                 spans[0][1] if spans else -1], dtype=torch.int64, device=dev)
            allc = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allc, mine)
            # true carry-in of this rank = carry-out of the nearest earlier rank that has a match
            cin = empty
            for r in range(rank):
                c = allc[r].tolist()
                if c[2]:
                    cin = (c[0], c[1], True)
            rerun = must_rerun(first_empty, used, cin)
            flag = torch.tensor([int(rerun)], dtype=torch.int64, device=dev)
            dist.all_reduce(flag)
            if rerun:
                spans = local_scan(own[0], own[1], *cin)
                used = cin
            if int(flag.item()) == 0:
                break
    total = len(spans)
    gathered = list(spans) if rank == 0 else None
    if world > 1:
        import torch

        dev = _device_for(dist)
        cnt = torch.tensor([len(spans)], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(counts, cnt)
        counts = [int(c.item()) for c in counts]
        total = sum(counts)
        if gather_to_root:
            # padded gather of (begin,end) pairs to rank 0: 16 B per match, direct peer->root
            width = max(max(counts), 1)
            buf = torch.zeros((width, 2), dtype=torch.int64, device=dev)
            if spans:
                buf[:len(spans)] = torch.tensor(spans, dtype=torch.int64, device=dev)
            parts = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
            dist.gather(buf, parts, dst=0)
            if rank == 0:
                gathered = []
                for r in range(world):
                    gathered += [tuple(x) for x in parts[r][:counts[r]].tolist()]
    return total, gathered, spans


def sharded_match_all_tensor(local_scan, ranges, rank: int, world: int, dist=None, device=None):
    """sharded_match_all on tensors: local_scan(lo, hi, carry_cur, carry_prev_end, have_prev) returns an
    (k, 2) int64 tensor of GLOBAL spans (any device); the carry exchange moves 5 integers per rank, the
    spans go to rank 0 through gather_spans_tensor.  Returns (total, spans on rank 0 or None, local)."""
    import torch

    own = ranges[rank]
    empty = (0, 0, False)
    spans = local_scan(own[0], own[1], *empty)
    if world == 1 or dist is None:
        return int(spans.shape[0]), spans, spans
    used = empty
    first_empty = tuple(spans[0].cpu().tolist()) if spans.shape[0] else None

    def head_tail(sp):
        if sp.shape[0] == 0:
            return None, None
        ht = torch.stack([sp[0], sp[-1]]).cpu().tolist()
        return tuple(ht[0]), tuple(ht[1])

    for _ in range(world + 1):
        first, last = head_tail(spans)
        co = carry_out([last] if last else [], used)
        mine = torch.tensor([co[0], co[1], int(co[2]), first[0] if first else -1, first[1] if first else -1],
                            dtype=torch.int64, device=device)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        cin = empty
        for r in range(rank):
            cc = allc[r].tolist()
            if cc[2]:
                cin = (cc[0], cc[1], True)
        rerun = must_rerun(first_empty, used, cin)
        flag = torch.tensor([int(rerun)], dtype=torch.int64, device=device)
        dist.all_reduce(flag)
        if rerun:
            spans = local_scan(own[0], own[1], *cin)
            used = cin
        if int(flag.item()) == 0:
            break
    gathered = gather_spans_tensor(spans.to(device), rank, world, dist)
    cnt = torch.tensor([int(spans.shape[0])], dtype=torch.int64, device=device)
    dist.all_reduce(cnt)
    return int(cnt.item()), gathered, spans


def gather_spans_tensor(spans, rank: int, world: int, dist, dst: int = 0):
    """Exchange step for MANY matches: `spans` is an (k, 2) int64 tensor on the collective device (HBM
    for RCCL); counts travel by all_gather, the pairs by one padded gather to `dst` -- no Python lists
    of tuples (C5's line tables are ~10^8 entries).  Returns the concatenated (K, 2) tensor on `dst`,
    None elsewhere."""
    import torch

    dev = spans.device
    cnt = torch.tensor([int(spans.shape[0])], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    buf = torch.zeros((width, 2), dtype=torch.int64, device=dev)
    if spans.shape[0]:
        buf[:spans.shape[0]] = spans
    parts = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, parts, dst=dst)
    if rank != dst:
        return None
    return torch.cat([parts[r][:counts[r]] for r in range(world)], dim=0)


def multi_pattern_counts(run_local, rerun_one, n_patterns: int, rank: int, world: int, dist, device):
    """Sharded COUNTS of several patterns over one text (regexdna's nine, BASELINE configs[2]) with the
    selection carried over the cuts.  run_local() -> (counts, bounds): this rank's counts with an empty
    carry and per pattern None or (first begin, first end, last begin, last end); rerun_one(i, carry_cur,
    carry_prev_end) -> (count, bounds_i) re-runs pattern i with the true carry.  One all_gather of
    8 integers per pattern per round -- count, the first match under the EMPTY carry, the current last match,
    and the carry the current result was selected under, so that every rank can tell which ranks re-run
    (must_rerun) --; a second round only when some rank's first match begins before its left neighbour's last
    end (self-overlapping occurrences across a cut: rare).  Returns the job-wide counts, identical on every
    rank."""
    import torch

    counts, bounds = run_local()
    first_empty = [None if b is None else (b[0], b[1]) for b in bounds]
    used = [(0, 0, False)] * n_patterns
    for _ in range(world + 1):
        mine = torch.full((n_patterns, 8), -1, dtype=torch.int64)
        for i in range(n_patterns):
            fe = first_empty[i] or (-1, -1)
            last = (bounds[i][2], bounds[i][3]) if bounds[i] is not None else (-1, -1)
            mine[i] = torch.tensor([counts[i], fe[0], fe[1], last[0], last[1], used[i][0], used[i][1], int(used[i][2])], dtype=torch.int64)
        mine = mine.to(device)
        allb = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allb, mine)
        allb = torch.stack(allb).cpu().tolist()   # [world][pattern][count, fb0, fe0, lb, le, used_cur, used_pe, used_have]
        again = False
        for i in range(n_patterns):
            # carry into rank r = the last match of the nearest rank before it that has one
            for r in range(1, world):
                prev = [q for q in range(r) if allb[q][i][3] >= 0]
                cin = (0, 0, False)
                if prev:
                    lb, le = allb[prev[-1]][i][3], allb[prev[-1]][i][4]
                    cin = (le if le > lb else lb + 1, le, True)
                row = allb[r][i]
                fe = (row[1], row[2]) if row[1] >= 0 else None
                if must_rerun(fe, (row[5], row[6], bool(row[7])), cin):
                    again = True
                    if r == rank:
                        counts[i], bounds[i] = rerun_one(i, cin[0], cin[1]) if cin[2] else _rerun_empty(rerun_one, i)
                        used[i] = cin
        if not again:
            return [sum(allb[r][i][0] for r in range(world)) for i in range(n_patterns)]
    raise RuntimeError("carry exchange did not converge")


class CarryExchange:
    """The same exchange with the rows left on the device (what bench.py's regexdna step uses under RCCL): per round
    ONE kernel that writes this rank's rows (rj_multi_bounds_device), ONE all_gather of (P, 8) integers, ONE kernel
    that takes the decision for this rank from the gathered rows (rj_carry_decide) into pinned host memory, and ONE
    synchronise -- no tensors built on the host, no per-pattern Python loop, two host syncs per step in all (the other
    one is rj_multi_run's own).  With a CPU process group (gloo: tests on one GPU) the rows take the detour over the
    host for the collective only."""

    def __init__(self, n_patterns: int, rank: int, world: int, dist, device, collective_device=None):
        import torch

        self.P, self.rank, self.world, self.dist = n_patterns, rank, world, dist
        self.device = device
        self.cdev = collective_device if collective_device is not None else device
        self.mine = torch.zeros((n_patterns, 8), dtype=torch.int64, device=device)
        self.all = torch.zeros((world, n_patterns, 8), dtype=torch.int64, device=device)
        self.out = torch.zeros(4 * n_patterns + 1, dtype=torch.int64).pin_memory()
        self.staged = self.cdev.type != device.type
        if self.staged:
            self.all_c = torch.zeros((world, n_patterns, 8), dtype=torch.int64, device=self.cdev)

    def counts(self, multi, run_local, rerun_one, offset: int, stream: int):
        """run_local() -> local counts (rj_multi_run over this rank's range, empty carry); rerun_one(i, cur, prev_end, have)
        re-runs pattern i of `multi` under the true carry (offsets LOCAL to the shard).  Returns the job-wide counts.
        `stream` must be torch's CURRENT stream of the device (the collective, the row update and the synchronise below
        run there; bench.py wraps the call in `torch.cuda.stream(...)`): a kernel queued on another stream would race
        with them."""
        import torch
        from . import api

        if stream != torch.cuda.current_stream(self.device).cuda_stream:
            raise ValueError("CarryExchange.counts: `stream` must be the device's current torch stream")
        run_local()
        first = True
        for _ in range(self.world + 1):
            multi.bounds_rows(self.mine, offset, first, stream)
            first = False
            if self.staged:
                self.dist.all_gather([self.all_c[r] for r in range(self.world)], self.mine.to(self.cdev))
                self.all.copy_(self.all_c)
            else:
                self.dist.all_gather_into_tensor(self.all.view(-1), self.mine.view(-1))
            api.carry_decide(self.all, self.world, self.rank, self.P, self.out, stream)
            torch.cuda.current_stream(self.device).synchronize()
            out = self.out.tolist()
            P = self.P
            if out[4 * P] == 0:
                return out[:P]
            for i in range(P):
                if out[P + i]:
                    cur, pe = out[2 * P + 2 * i], out[2 * P + 2 * i + 1]
                    have = cur != 0 or pe != 0
                    # in the shard's own coordinates; a previous match that ends before the buffer begins cannot touch
                    # anything in it (clamping its end to 0 would suppress a legitimate empty match at local 0): no carry
                    have_local = have and pe >= offset
                    rerun_one(i, max(cur - offset, 0) if have_local else 0, pe - offset if have_local else 0, have_local)
                    self.mine[i, 5:8] = torch.tensor([cur, pe, int(have)], dtype=torch.int64)
        raise RuntimeError("carry exchange did not converge")


def _rerun_empty(rerun_one, i):
    # (a carry that went back to "no earlier match": select again from an empty carry -- have = False when the callback
    # takes the flag)
    import inspect
    try:
        n_params = len(inspect.signature(rerun_one).parameters)
    except (TypeError, ValueError):     # (a callable without a signature: take the current form)
        n_params = 4
    return rerun_one(i, 0, 0, False) if n_params >= 4 else rerun_one(i, 0, 0)


def _device_for(dist):
    import torch

    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


# ----------------------------------------------------------------------------- many files (jrep, C5)
def partition_files(sizes, world):
    """Greedy bin packing of files onto ranks by size (largest first, least-loaded rank), the file
    sharding of a grep over a tree: every rank walks the same tree, so the assignment is computed
    identically everywhere and nothing is exchanged.  Returns, per rank, the indices of its files in
    their original order."""
    load = [0] * world
    owner = [0] * len(sizes)
    for i in sorted(range(len(sizes)), key=lambda k: (-sizes[k], k)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += sizes[i] + 1
    return [[i for i in range(len(sizes)) if owner[i] == r] for r in range(world)]


def gather_bytes(blob: bytes, rank, world, dist, device=None):
    """Exchange step of the file-sharded path: every rank's output (bytes) on rank 0, in rank order
    (other ranks get None).  Two collectives over RCCL/gloo: all_gather of the lengths, all_gather of
    the padded byte tensors."""
    import torch
    if world == 1:
        return blob
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    width = max(max(lens), 1)
    mine = torch.zeros(width, dtype=torch.uint8, device=dev)
    if blob:
        mine[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    parts = [torch.zeros(width, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank != 0:
        return None
    return b"".join(bytes(p[:k].cpu().numpy().tobytes()) for p, k in zip(parts, lens))
