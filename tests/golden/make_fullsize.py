#!/usr/bin/env python3
"""Generates tests/golden/fullsize_vectors.json: the REAL reference's answers at BASELINE's full sizes, as counts
and span digests (rejit_amd.workloads.span_digest_numpy) -- run ONLY in the build container, where /root/reference
exists and oracle/_ref has been built, and needs ~20 GB of RAM and a few minutes:

    make -C oracle ref && python tests/golden/make_fullsize.py [c3] [c2] [c4] [c4all] [c5]

  C3  regexdna: the nine patterns over the stripped 50M-line FASTA (500 000 000 bytes, rejit_amd.workloads), reference
      with use_fast_forward=1 / use_ff_reduce=0 (its correct fast configuration) AND with
      use_fast_forward=0 (the oracle configuration): both must agree.
  C2  literal `regexp` over 5 000 000 000 bytes of the seeded random ASCII stream with bench.py's 1000 planted
      occurrences; fast configuration and use_fast_forward=0.
  C4  the complex benchmark regex over rank 0's shard of bench.py --workload complex --gpus 8 --literal-bytes 6250000000:
      the bytes [0, 6 250 000 057) of the stream with the job's planted samples, matches that BEGIN before the cut;
      use_fast_forward=0 (the fast path mis-places matches of this regex).

  C4ALL every one of the eight shards of that job (round 4), each with 64 KiB of left context: the matches that begin in
      the shard's own range, as global offsets; use_fast_forward=0.
  C5  bench.py's jrep_10gb tree (100 000 files, 10 GB): per file the reference's MatchAll count of `regexp` and, for the
      files with a match, of `^` -- what sample/jrep.cc:288-294 computes -- as totals and a sha256 of the rows.

What is stored is data: the inputs' generator parameters and the expected counts / digests."""
import ctypes
import json
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from checkers import Ref  # noqa: E402
from rejit_amd import sharding, workloads as W  # noqa: E402

OUT = os.path.join(HERE, "fullsize_vectors.json")
FAST, FF0 = (1, 1, 0, 1), (0, 0, 1, 1)   # (use_fast_forward, use_fast_forward_early, use_ff_reduce, use_parser_opt)


def ref_spans(ref, flags, rx: bytes, text: np.ndarray, cap: int = 1 << 24) -> np.ndarray:
    ref.set_flags(*flags)
    buf = np.zeros(2 * cap, dtype=np.uint64)
    n = ref.lib.ref_match_all(rx, text.ctypes.data_as(ctypes.c_char_p), len(text), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cap)
    assert 0 <= n <= cap, n
    return buf[:2 * n].reshape(-1, 2).copy()


def random_ascii_big(n, seed, start=0, chunk=1 << 27):
    out = np.empty(n, dtype=np.uint8)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        out[a:b] = W.random_ascii_numpy(b - a, seed, start=start + a)
    return out


def _c4_shard(job):
    """One shard of the 8-GPU C4 job in its own process (the reference library keeps global flags: one instance per process)."""
    r, world, per, left = job
    ref = Ref(use_ff=0)
    ref.lib.ref_match_all.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t]
    rx = W.BENCH_REGEXES[3][0]
    n_total = per * world
    ranges = sharding.partition(n_total, world)
    cuts = [q[0] for q in ranges[1:]]
    rng = random.Random(7)
    needles = [(o, W.complex_regex_sample(rng)) for o in W.plant_offsets(n_total, 64, 200 * world, seed=7, boundaries=cuts)]
    own = ranges[r]
    lo = max(0, own[0] - left)
    hi = min(n_total, own[1] + 58)
    t0 = time.time()
    text = random_ascii_big(hi - lo, 0xC0FFEE, start=lo)
    for o, smp in needles:
        a, b = max(o, lo), min(o + len(smp), hi)
        if a < b:
            W.plant(text, [a - lo], smp[a - o:b - o])
    t1 = time.time()
    sp = ref_spans(ref, FF0, rx.encode(), text) + np.uint64(lo)
    before = sp[sp[:, 0] < np.uint64(own[0])]
    if r > 0:
        # a gap without a match between the fresh start and the own range: both runs agree from there on
        edges = [lo] + [int(e) for e in before[:, 1]]
        starts = [int(b) for b in before[:, 0]] + [own[0]]
        assert any(s_ - e_ > 64 for e_, s_ in zip(edges, starts)), "no gap in the left context of shard %d" % r
    mine = sp[(sp[:, 0] >= np.uint64(own[0])) & (sp[:, 0] < np.uint64(min(own[1], n_total + 1)))]
    d = W.span_digest_numpy(mine)
    carry = [int(before[-1, 0]), int(before[-1, 1])] if len(before) else None
    print("  shard %d done: text %.0f s, reference (ff off) %.0f s" % (r, t1 - t0, time.time() - t1), flush=True)
    return {"rank": r, "own": [int(own[0]), int(min(own[1], n_total + 1))], "digest": d, "last_match_before": carry}


def main():
    which = set(sys.argv[1:]) or {"c3", "c2", "c4"}
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {}
    ref = Ref(use_ff=0)
    ref.lib.ref_match_all.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t]
    if "c3" in which:
        nf = 50_000_000
        t0 = time.time()
        text = W.fasta_stripped_numpy(nf)
        print("C3 text: %d bytes in %.1f s" % (len(text), time.time() - t0), flush=True)
        rows = []
        for rx in W.REGEXDNA_PATTERNS:
            t0 = time.time()
            fast = W.span_digest_numpy(ref_spans(ref, FAST, rx.encode(), text))
            t1 = time.time()
            slow = W.span_digest_numpy(ref_spans(ref, FF0, rx.encode(), text))
            assert fast == slow, (rx, fast, slow)
            print("  %-28s %8d matches  (ff on %.1f s, ff off %.1f s)" % (rx, fast["count"], t1 - t0, time.time() - t1), flush=True)
            rows.append({"regex": rx, "digest": fast})
        doc["c3"] = {"text": "rejit_amd.workloads.fasta_stripped(50_000_000): 500 000 000 bytes", "fasta_n": nf, "bytes": int(len(text)),
                     "reference_flags": "use_fast_forward=1,use_ff_reduce=0 and use_fast_forward=0: identical", "patterns": rows}
        del text
    if "c2" in which:
        n, seed = 5_000_000_000, 0xC0FFEE
        t0 = time.time()
        text = random_ascii_big(n, seed)
        offs = W.plant_offsets(n, 6, 1000, seed=seed, boundaries=[16, 1024, 1 << 20, 1 << 30, n // 2])
        W.plant(text, offs, b"regexp")
        print("C2 text: %d bytes in %.1f s" % (n, time.time() - t0), flush=True)
        t0 = time.time()
        fast = W.span_digest_numpy(ref_spans(ref, FAST, b"regexp", text))
        t1 = time.time()
        slow = W.span_digest_numpy(ref_spans(ref, FF0, b"regexp", text))
        assert fast == slow, (fast, slow)
        print("  regexp: %d matches (%d planted)  (ff on %.1f s, ff off %.1f s)" % (fast["count"], len(offs), t1 - t0, time.time() - t1), flush=True)
        doc["c2"] = {"text": "random_ascii(5e9, seed 0xC0FFEE) + plant_offsets(n, 6, 1000, seed, [16, 1024, 2^20, 2^30, n/2]) x 'regexp'",
                     "bytes": n, "seed": seed, "planted": len(offs), "regex": "regexp",
                     "reference_flags": "use_fast_forward=1,use_ff_reduce=0 and use_fast_forward=0: identical", "digest": fast}
        del text
    if "c4" in which:
        world, per = 8, 6_250_000_000
        rx = W.BENCH_REGEXES[3][0]
        n_total = per * world
        ranges = sharding.partition(n_total, world)
        own = ranges[0]
        vis_lo, vis_hi = sharding.visible_range(n_total, own, 58)
        assert vis_lo == 0
        t0 = time.time()
        text = random_ascii_big(vis_hi, 0xC0FFEE)
        cuts = [r[0] for r in ranges[1:]]
        rng = random.Random(7)
        needles = [(o, W.complex_regex_sample(rng)) for o in W.plant_offsets(n_total, 64, 200 * world, seed=7, boundaries=cuts)]
        planted = 0
        for o, s in needles:
            lo, hi = max(o, vis_lo), min(o + len(s), vis_hi)
            if lo < hi:
                W.plant(text, [lo], s[lo - o:hi - o])
                planted += 1
        print("C4 shard: %d bytes (%d samples inside) in %.1f s" % (vis_hi, planted, time.time() - t0), flush=True)
        t0 = time.time()
        sp = ref_spans(ref, FF0, rx.encode(), text)
        sp = sp[sp[:, 0] < np.uint64(own[1])]
        d = W.span_digest_numpy(sp)
        print("  complex: %d matches begin in the shard (ff off %.1f s)" % (d["count"], time.time() - t0), flush=True)
        doc["c4"] = {"text": "rank 0 of bench.py --workload complex --gpus 8 --literal-bytes 6250000000: bytes [0, %d) of random_ascii(seed 0xC0FFEE) "
                             "with the job's planted complex_regex_sample strings" % vis_hi,
                     "world": world, "bytes_per_gpu": per, "visible_bytes": int(vis_hi), "own_end": int(own[1]), "regex": rx,
                     "reference_flags": "use_fast_forward=0", "digest": d}
        del text
    if "c4all" in which:
        # Every shard of the 8-GPU job (BASELINE configs[3]: 50 GB), one after the other: the reference over the shard's
        # bytes plus 64 KiB of left context (a fresh run there agrees with the run over the whole text from the first gap
        # without a match on -- asserted), matches that BEGIN in the shard's own range, as global offsets.  The digests
        # are sums: the whole job's digest is the sum of the shards'.
        world, per = 8, 6_250_000_000
        rx = W.BENCH_REGEXES[3][0]
        n_total = per * world
        ranges = sharding.partition(n_total, world)
        cuts = [r[0] for r in ranges[1:]]
        rng = random.Random(7)
        needles = [(o, W.complex_regex_sample(rng)) for o in W.plant_offsets(n_total, 64, 200 * world, seed=7, boundaries=cuts)]
        left = 1 << 16
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=int(os.environ.get("C4ALL_WORKERS", "4"))) as pool:
            shards = list(pool.map(_c4_shard, [(r, world, per, left) for r in range(world)]))
        for x in shards:
            print("  shard %d: bytes [%d, %d), %d matches begin in it" % (x["rank"], x["own"][0], x["own"][1], x["digest"]["count"]), flush=True)
        doc["c4all"] = {"text": "bench.py --workload complex --gpus 8 --literal-bytes 6250000000: random_ascii(seed 0xC0FFEE) with the job's planted "
                                "complex_regex_sample strings; per shard the matches that begin in its own range, global offsets",
                        "world": world, "bytes_per_gpu": per, "regex": rx, "reference_flags": "use_fast_forward=0", "shards": shards,
                        "total_count": sum(x["digest"]["count"] for x in shards)}
    if "c5" in which:
        # BASELINE configs[4] at its size: bench.py's jrep_10gb tree (100 000 files, 10 GB), the reference's own jrep logic
        # (sample/jrep.cc:288-294: MatchAll per file, then `^` MatchAll of every file with a match), per file
        import bench
        n_files, total_bytes = 100_000, 10_000_000_000
        t0 = time.time()
        files = bench.jrep_tree(n_files, total_bytes)
        print("C5 tree: %d files, %d bytes in %.0f s" % (n_files, sum(len(f) for f in files), time.time() - t0), flush=True)
        ref.set_flags(*FAST)
        t0 = time.time()
        rows = []
        for i, f in enumerate(files):
            k = int(ref.lib.ref_match_all_repeat(b"regexp", f, len(f), 1))
            if k:
                l = int(ref.lib.ref_match_all_repeat(b"^", f, len(f), 1))
                rows.append((i, k, l))
        print("  %d files with matches, %d matches, %d line starts (reference, %.0f s)" % (len(rows), sum(r[1] for r in rows), sum(r[2] for r in rows), time.time() - t0), flush=True)
        # cross-check of the literal with Python on a sample of the files
        for i in range(0, n_files, 997):
            assert (files[i].count(b"regexp") > 0) == any(r[0] == i for r in rows), i
        doc["c5"] = {"text": "bench.jrep_tree(100000, 10000000000): log-normal file sizes, ` regexp ` in every hundredth file", "files": n_files,
                     "bytes_asked": total_bytes, "bytes": int(sum(len(f) for f in files)), "regex": "regexp", "line_regex": "^",
                     "reference_flags": "use_fast_forward=1,use_ff_reduce=0", "files_with_matches": len(rows), "matches": int(sum(r[1] for r in rows)),
                     "line_starts": int(sum(r[2] for r in rows)), "sha256": bench.jrep_digest(rows)}
        del files
    with open(OUT, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
