#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json: "GB/s text scanned + matches/s,
regexdna 50M-line input, 1/2/4/8 MI355X").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path over one batch of synthetic input: the nine regex-dna
patterns (reference sample/regexdna.cc:51-67), one MatchAllCount each, over the stripped
50M-line FASTA text (500 MB per GPU), already resident in HBM when the timed region
starts.  With N GPUs the text is N x 500 MB (weak scaling), cut into contiguous byte ranges
with a halo of max_match_len-1 bytes; the only exchange is an RCCL all_reduce of the nine
match counts per step.  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      -- the dominant kernel (the fast-forward window scan): algorithmic bytes
                   per launch (1 byte read per text byte per MatchAll call, SURVEY.md
                   section 8d) / average launch duration from HIP events on the run's stream
  cpu_baseline  -- the REAL reference (oracle/_ref, built from /root/reference) timed on
                   one host core on a bounded sample of the same text, same convention
  literal_scan  -- BASELINE configs[1]: literal `regexp` over 5 GB random ASCII, 1 GPU
                   (the pure fast-forward scan of the north star), with its own roofline
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(key, **match):
    """HBM bytes per launch of the dominant kernel, measured with rocprofv3 --pmc FETCH_SIZE in a
    separate pass and corrected as the guide prescribes (x2 on gfx950); committed under
    profiles/.  Reported only when the profiled configuration is the one being run, else null."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[key]
        if all(rec.get(k) == v for k, v in match.items()):
            return int(rec["hbm_read_bytes_per_launch"])
    except Exception:
        pass
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fasta-n", type=int, default=50_000_000, help="FASTA size parameter per GPU (50M = 500 MB stripped)")
    ap.add_argument("--literal-bytes", type=int, default=5_000_000_000)
    ap.add_argument("--no-extra", action="store_true", help="skip the literal_scan extra")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-calls", action="store_true", help="headline with nine synchronous rj_scan_run calls per step "
                    "instead of one rj_multi_run (separate scans, batched tails)")
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for "
                    "functional tests of the multi-rank path on a 1-GPU box together with --same-device")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--pattern-threads", type=int, default=1, help="host threads (one HIP stream each) issuing the 9 "
                    "MatchAllCount calls of a step, as sample/regexdna-multithread.cc does; 1 = strictly sequential")
    return ap.parse_args()


def cpu_baseline(text_host: bytes, patterns, sample_desc: str):
    """The reference's own x86 SIMD path on ONE host core (kind "reference"), or, when the
    prebuilt oracle/_ref library is absent, our C restatement (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import checkers
    n = len(text_host)
    if checkers.have_ref():
        # default flags except use_ff_reduce=0: the fast-forward configuration whose counts are
        # correct; compile time excluded as in the reference harness
        ref = checkers.Ref(use_ff=1, ff_early=1, ff_reduce=0, parser_opt=1)
        counts = []
        t0 = time.perf_counter()
        for rx in patterns:
            counts.append(int(ref.lib.ref_match_all_repeat(rx.encode(), text_host, n, 1)))
        dt = time.perf_counter() - t0
        return dict(value=len(patterns) * n / dt / 1e9, unit="GB/s", cores=1, kind="reference",
                    sample=sample_desc + "; reference flags use_fast_forward=1 use_ff_reduce=0",
                    seconds=round(dt, 3)), counts
    oracle = checkers.Oracle()
    n = min(n, 4 << 20)
    t0 = time.perf_counter()
    counts = [oracle.count(rx.encode(), text_host[:n]) for rx in patterns]
    dt = time.perf_counter() - t0
    return dict(value=len(patterns) * n / dt / 1e9, unit="GB/s", cores=1, kind="port",
                sample=f"first {n} bytes of rank 0's text (oracle/_ref not present)", seconds=round(dt, 3)), None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import rejit_amd
    from rejit_amd import sharding, workloads as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collective buffers live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    rejit_amd.build()
    patterns = W.REGEXDNA_PATTERNS
    progs = [rejit_amd.Program(rx) for rx in patterns]
    scans = [rejit_amd.Scan(p) for p in progs]
    max_len = max(p.info()["max_len"] for p in progs)

    # ---- input: N x (10 * fasta_n) bytes of stripped FASTA, rank r holds its range + halo
    n_fa = args.fasta_n * world
    n_total = W.fasta_stripped_size(n_fa)
    ranges = sharding.partition(n_total, world)
    own = ranges[rank]
    vis_lo, vis_hi = sharding.visible_range(n_total, own, max_len)
    text = W.fasta_stripped_torch(n_fa, dev, lo=vis_lo, hi=vis_hi)
    n_local = int(text.numel())
    own_lo, own_hi = own[0] - vis_lo, min(own[1], n_total + 1) - vis_lo
    own_bytes = min(own[1], n_total) - own[0]
    stream = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize(dev)

    scan_ms = []
    # The nine patterns of a step are independent calls: like the reference's own
    # sample/regexdna-multithread.cc:65-78 they are issued from a few host threads, each on its
    # own HIP stream, so one call's host-side latency (launches, the result read-back) hides
    # under the others' kernels.  The kernels themselves still share the one GPU.
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, min(args.pattern_threads, len(scans)))
    streams = [torch.cuda.Stream(device=dev) for _ in scans]
    pool = ThreadPoolExecutor(max_workers=n_thr) if n_thr > 1 else None
    text_ptr = text.data_ptr()

    def run_one(i, own_stream=None):
        torch.cuda.set_device(dev)
        st_i = streams[i].cuda_stream if (pool or own_stream) else stream
        return scans[i].run(text_ptr, n_local, own_begin=own_lo, own_end=own_hi, stream=st_i)

    # Headline: the nine patterns of a step go through rj_multi_run in its "separate
    # scans" mode -- nine scan kernels queued back to back (each the ordinary single-pattern kernel at
    # its full streaming rate, timed by its own dispatch timestamps: the roofline below), then the
    # verify / gather tails of all nine patterns in two launches and ONE host synchronise, instead of
    # nine round trips.  (The fused single-kernel mode is reported as `fused`, the nine synchronous
    # calls as `serial_calls`.)
    use_multi = not pool and not args.serial_calls
    multi_sep = None
    if use_multi:
        multi_sep = rejit_amd.MultiScan(progs)
        multi_sep.set_mode(1)
        sep_scans = [multi_sep.scan(i) for i in range(len(progs))]

    pending = []   # (all_reduce work, device tensor) of steps whose exchange is still in flight

    def step(record: bool):
        if pool:
            local = list(pool.map(run_one, range(len(scans))))
        elif use_multi:
            local = multi_sep.run(text_ptr, n_local, stream=stream, own_begin=own_lo, own_end=own_hi)
        else:
            local = [run_one(i) for i in range(len(scans))]
        if record:
            scan_ms.extend(sc.stats()["scan_ms"] for sc in (sep_scans if use_multi else scans))
        if world > 1:
            # exchange step of the path: sum of the 9 match counts (72 bytes) over the ranks, RCCL over
            # xGMI.  It is issued asynchronously and collected one step later, so the next step's scans
            # run while the (latency-only) all_reduce is in flight.
            t = torch.tensor(local, dtype=torch.int64).to(cdev)
            pending.append((dist.all_reduce(t, async_op=True), t))
            while len(pending) > 1:
                w, _ = pending.pop(0)
                w.wait()
            return None
        return local

    def drain():
        """Complete the outstanding exchanges; returns the job-wide counts of the last step."""
        last = None
        while pending:
            w, t = pending.pop(0)
            w.wait()
            last = t
        return last.tolist() if last is not None else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts = step(True)
    if world > 1:
        counts = drain()          # inside the timed region: every step's exchange has completed
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_matches = int(sum(counts))
    scanned = len(patterns) * n_total * args.steps          # bytes of text scanned by the whole job
    value = scanned / elapsed / 1e9
    avg_scan_ms = sum(scan_ms) / max(len(scan_ms), 1)
    achieved = own_bytes / (avg_scan_ms * 1e-3) / 1e9 if avg_scan_ms > 0 else 0.0

    out = {
        "metric": "GB/s text scanned (regexdna 9 patterns, 50M-line input); matches/s alongside",
        "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "regexdna: 9 x MatchAllCount over the stripped 50M-line FASTA (BASELINE configs[2])",
                   "fasta_n_per_gpu": args.fasta_n, "text_bytes_per_gpu": int(own_bytes), "patterns": len(patterns),
                   "sharding": "contiguous byte ranges + %d-byte halo; all_reduce of 9 counts per step" % (max_len - 1),
                   "pattern_threads": n_thr,
                   "calls": "rj_multi_run, separate scan kernels + batched tails (mode 1)" if use_multi else "9 x rj_scan_run per step"},
        "matches_per_s": round(total_matches * args.steps / elapsed, 1),
        "matches_per_pass": counts,
        "roofline": {"bound": "hbm", "kernel": "scan_windows<K>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": pmc_traffic("regexdna", fasta_n=args.fasta_n) if world == 1 else None,
                     "avg_launch_ms": round(avg_scan_ms, 5), "bytes_per_launch": int(own_bytes),
                     "launches_timed": len(scan_ms)},
    }

    if rank == 0 and world == 1 and not args.no_extra and use_multi:
        # the same job as nine synchronous calls per step (every call waits for its own tail)
        for _ in range(2):
            cs = [run_one(i) for i in range(len(scans))]
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            cs = [run_one(i) for i in range(len(scans))]
        torch.cuda.synchronize(dev)
        es = time.perf_counter() - t1
        assert cs == counts, "synchronous calls disagree with rj_multi_run"
        out["serial_calls"] = {"calls": "9 x rj_scan_run per step", "value": round(scanned / es / 1e9, 3), "unit": "GB/s",
                               "ms_per_step": round(es / args.steps * 1e3, 4)}

    if rank == 0 and world == 1 and not args.no_extra and use_multi:
        # the same with the scan kernels alternating between two streams (rj_multi mode 2): consecutive
        # kernels overlap at their boundaries instead of draining and ramping up one by one.  Faster,
        # but two scan kernels then share the GPU part of the time, so a per-kernel duration (rocprof
        # shows ~160 us each) is no roofline input any more -- which is why it is not the headline.
        multi_il = rejit_amd.MultiScan(progs)
        multi_il.set_mode(2)
        for _ in range(2):
            ci = multi_il.run(text_ptr, n_local, stream=stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ci = multi_il.run(text_ptr, n_local, stream=stream)
        torch.cuda.synchronize(dev)
        ei = time.perf_counter() - t1
        assert ci == counts, "interleaved run disagrees"
        out["interleaved"] = {"calls": "rj_multi_run mode 2 (scan kernels on two alternating streams)",
                              "value": round(scanned / ei / 1e9, 3), "unit": "GB/s", "ms_per_step": round(ei / args.steps * 1e3, 4)}

    if rank == 0 and world == 1 and not args.no_extra and n_thr == 1:
        # the same job with the 9 calls issued from 3 host threads (one stream each): the calls'
        # host-side latency overlaps, the kernels share the GPU (so per-kernel times stretch, which
        # is why the headline run above -- the one the roofline is taken from -- stays serial)
        with ThreadPoolExecutor(max_workers=3) as pool3:
            for _ in range(2):
                list(pool3.map(lambda i: run_one(i, True), range(len(scans))))
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                c3 = list(pool3.map(lambda i: run_one(i, True), range(len(scans))))
            torch.cuda.synchronize(dev)
            e3 = time.perf_counter() - t1
        assert c3 == counts, "threaded run disagrees with the serial run"
        out["overlapped"] = {"pattern_threads": 3, "value": round(scanned / e3 / 1e9, 3), "unit": "GB/s",
                             "ms_per_step": round(e3 / args.steps * 1e3, 4)}

    if rank == 0 and world == 1 and not args.no_extra:
        # the nine patterns in ONE pass over the text (rj_multi).  Same results; the
        # text is read once instead of nine times, and the kernel becomes VALU-bound.
        multi = rejit_amd.MultiScan(progs)
        for _ in range(2):
            cf = multi.run(text_ptr, n_local, stream=stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        fms = []
        for _ in range(args.steps):
            cf = multi.run(text_ptr, n_local, stream=stream)
            fms.append(multi.scan_ms())
        torch.cuda.synchronize(dev)
        ef = time.perf_counter() - t1
        assert cf == counts, "fused run disagrees with the nine single runs"
        a_ms = sum(fms) / len(fms)
        out["fused"] = {"api": "rj_multi_run (9 patterns, one pass over the text)", "fused": bool(multi.fused),
                        "value": round(scanned / ef / 1e9, 3), "unit": "GB/s", "ms_per_step": round(ef / args.steps * 1e3, 4),
                        "scan_kernel_ms": round(a_ms, 5),
                        "hbm_read_GBps": round(own_bytes / (a_ms * 1e-3) / 1e9, 1) if a_ms > 0 else None,
                        "bound": "VALU issue (~29 ops per text byte for 18 windows), not HBM"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded sample taken from the lower-case (matching) part of the text: the first 20 %
        # is the upper-case ALU repeat, which no pattern can match
        sample = min(n_local, args.cpu_sample_mib << 20)
        s0 = min(max(0, n_local - sample), (int(n_local * 0.6) // 4096) * 4096)
        host = text[s0:s0 + sample].cpu().numpy().tobytes()
        base, ref_counts = cpu_baseline(host, patterns, f"bytes [{s0}, {s0 + sample}) of the same text, 9 patterns, 1 pass")
        out["cpu_baseline"] = base
        if ref_counts is not None:
            # in-run parity check against the real reference on the sample
            gpu_counts = []
            for sc in scans:
                sc.run(text.data_ptr() + s0, sample, own_begin=0, own_end=sample + 1, stream=stream)
                gpu_counts.append(sc.stats()["n_matches"])
            assert gpu_counts == ref_counts, ("GPU and reference disagree on the sample", gpu_counts, ref_counts)
            out["cpu_baseline"]["parity_on_sample"] = "GPU counts == reference counts: %s" % ref_counts

    if rank == 0 and world == 1 and not args.no_extra:
        del text
        torch.cuda.empty_cache()
        n = args.literal_bytes
        t = W.random_ascii_torch(n, 0xC0FFEE, dev)
        offs = W.plant_offsets(n, 6, 1000, seed=0xC0FFEE, boundaries=[16, 1024, 1 << 20, 1 << 30, n // 2])
        W.plant(t, offs, b"regexp")
        prog = rejit_amd.Program("regexp")
        sc = rejit_amd.Scan(prog)
        torch.cuda.synchronize(dev)
        for _ in range(2):
            sc.run(t.data_ptr(), n, stream=stream)
        ms, tot = [], []
        steps = max(5, min(args.steps, 20))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            c = sc.run(t.data_ptr(), n, stream=stream)
            st = sc.stats()
            ms.append(st["scan_ms"])
            tot.append(st["total_ms"])
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        found = [b for b, _ in sc.spans()]
        assert set(offs) <= set(found), "a planted occurrence was missed"
        a_ms = sum(ms) / len(ms)
        ach = n / (a_ms * 1e-3) / 1e9
        out["literal_scan"] = {
            "workload": "literal 'regexp' MatchAll over %d bytes random ASCII ['0','z'), %d planted (BASELINE configs[1])" % (n, len(offs)),
            "value": round(n * steps / dt / 1e9, 1), "unit": "GB/s", "matches": int(c), "latency_ms": round(dt / steps * 1e3, 4),
            "roofline": {"bound": "hbm", "kernel": "scan_windows<1>", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("literal", bytes=n),
                         "avg_launch_ms": round(a_ms, 5), "bytes_per_launch": n},
        }

    if rank == 0 and world == 1 and not args.no_extra:
        # BASELINE configs[3] shape on one GPU: the complex benchmark regex (floating fast-forward
        # window `abcdefgh`) over the same 5 GB text with strings of its language planted
        import random as _random
        rx = W.BENCH_REGEXES[3][0]
        rng = _random.Random(7)
        offs2 = W.plant_offsets(n, 80, 1000, seed=7)
        samples = [W.complex_regex_sample(rng) for _ in offs2]
        for o, smp in zip(offs2, samples):
            W.plant(t, [o + 8], smp)
        sc2 = rejit_amd.Scan(rejit_amd.Program(rx))
        for _ in range(2):
            sc2.run(t.data_ptr(), n, stream=stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ms2 = []
        for _ in range(5):
            c2 = sc2.run(t.data_ptr(), n, stream=stream)
            ms2.append(sc2.stats()["scan_ms"])
        torch.cuda.synchronize(dev)
        dt2 = time.perf_counter() - t0
        ends = {e for _, e in sc2.spans()}
        assert all(o + 8 + len(smp) in ends for o, smp in zip(offs2, samples)), "a planted complex match was missed"
        a2 = sum(ms2) / len(ms2)
        out["complex_scan"] = {
            "workload": "%s MatchAll over %d bytes random ASCII, %d planted (BASELINE configs[3] shape, 1 GPU)" % (rx, n, len(offs2)),
            "value": round(n * 5 / dt2 / 1e9, 1), "unit": "GB/s", "matches": int(c2), "latency_ms": round(dt2 / 5 * 1e3, 4),
            "roofline": {"bound": "hbm", "kernel": "scan_windows<1> (floating window)", "achieved": round(n / (a2 * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(n / (a2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": None, "avg_launch_ms": round(a2, 5), "bytes_per_launch": n},
        }

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
