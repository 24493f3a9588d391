"""Probe of MatchAllCount in one kernel (plane_count.hip, rj_multi_set_counts_only) against the span pipeline
(plane_scan + classify_shared_multi + offsets_gather_check_multi) on the stripped FASTA text: synchronous step, the
two-in-flight loop of bench.py, the scan kernel's own duration.  A/B within ONE process / gpurun call.
    python tools/count_probe.py [fasta_n] [steps] [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import rejit_amd
from rejit_amd import workloads as W

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
text = W.fasta_stripped_torch(nf, dev)
n = int(text.numel())
progs = [rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS]


def sync_loop(counts_only):
    m = rejit_amd.MultiScan(progs)
    took = m.set_counts_only(counts_only)
    c = m.run(text.data_ptr(), n, stream=st)
    for _ in range(5):
        m.run(text.data_ptr(), n, stream=st)
    torch.cuda.synchronize(dev)
    ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        m.run(text.data_ptr(), n, stream=st)
        ms.append(m.scan_ms())
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    k = sum(ms) / len(ms)
    return c, m.how, dt, k, min(ms), took


def flight_loop(counts_only, tail_streams, timed_all=False):
    ms = [rejit_amd.MultiScan(progs) for _ in range(2)]
    for i, m in enumerate(ms):
        m.set_counts_only(counts_only)
        m.set_timing(i == 0 or timed_all)
        if tail_streams and not counts_only:
            m.set_tail_stream(True)
    kt = []
    c = None

    def loop(k, record):
        nonlocal c
        busy = [False, False]
        for i in range(k):
            j = i % 2
            if busy[j]:
                c = ms[j].finish()
                if record and (j == 0 or timed_all):
                    kt.append(ms[j].scan_ms())
            ms[j].start(text.data_ptr(), n, stream=st)
            busy[j] = True
        for j in ((k % 2), ((k + 1) % 2)):
            if busy[j]:
                c = ms[j].finish()
    loop(steps, False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    loop(steps, True)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    return c, dt, (sum(kt) / max(len(kt), 1))


print(f"text {n} bytes, {steps} steps per loop", flush=True)
for r in range(rounds):
    for co in (False, True):
        c, how, dt, k, kmin, took = sync_loop(co)
        print(f"round {r} counts_only={int(co)} how={how}: synchronous step {dt * 1e3:.4f} ms, scan kernel {k:.4f} ms (min {kmin:.4f}) = "
              f"{n / k / 1e6 / 8000:.3f} of peak", flush=True)
        cf, dtf, kf = flight_loop(co, True)
        print(f"round {r} counts_only={int(co)}: two in flight {dtf * 1e3:.4f} ms/step ({n / dtf / 1e9 / 8000:.3f} of peak), scan kernel in the loop {kf:.4f} ms = "
              f"{n / max(kf, 1e-9) / 1e6 / 8000:.3f}", flush=True)
        assert cf == c, (cf, c)
        if co:
            counts_c = c
        else:
            counts_s = c
    assert counts_c == counts_s, (counts_c, counts_s)
print("counts", counts_c)
