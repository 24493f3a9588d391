#!/usr/bin/env python3
"""First thing to run on the GPU in round 4: rj_multi_set_tail_stream (written at the end of round 3 without a GPU).
Counts of the nine regexdna patterns with two steps in flight, three ways -- both objects on one stream (the round-3
headline), a stream per object with ordered scans (`overlapped_tails`), one stream + the tails on a stream per object
(`tails_on_own_streams`) -- each checked against a synchronous run, on a small and on the 500 MB text, with step times.
usage: tail_stream_probe.py [fasta_n, default 50000000] [steps, default 40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rejit_amd
from rejit_amd import workloads as W

fasta_n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
progs = [rejit_amd.Program(p) for p in W.REGEXDNA_PATTERNS]
main = torch.cuda.current_stream(dev).cuda_stream


def loop(text, n, variant):
    ms = [rejit_amd.MultiScan(progs), rejit_amd.MultiScan(progs)]
    second = torch.cuda.Stream(dev)
    streams = [main, main]
    if variant == "stream_per_object":
        streams = [main, second.cuda_stream]
        ms[0].order_after(ms[1]); ms[1].order_after(ms[0])
    if variant == "tails_on_own_streams":
        for m in ms:
            m.set_tail_stream(True)
    out, busy = [], [False, False]
    for m in ms:                       # warm: region sizes, table uploads
        m.run(text.data_ptr(), n, stream=main)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        j = k & 1
        if busy[j]:
            out.append(ms[j].finish())
        ms[j].start(text.data_ptr(), n, stream=streams[j])
        busy[j] = True
    for j in ((steps & 1), (steps + 1) & 1):
        if busy[j]:
            out.append(ms[j].finish())
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    scan = sum(m.scan_ms() for m in ms) / 2
    return out, dt, scan


for nf in sorted({200_000, fasta_n}):
    text = W.fasta_stripped_torch(nf, dev)
    n = int(text.numel())
    want = rejit_amd.MultiScan(progs).run(text.data_ptr(), n, stream=main)
    for variant in ("one_stream", "stream_per_object", "tails_on_own_streams"):
        got, dt, scan = loop(text, n, variant)
        ok = all(g == want for g in got) and len(got) == steps
        print(f"fasta_n {nf:9d} {variant:22s} {'OK ' if ok else 'WRONG'} {dt * 1e3:8.4f} ms/step  {9 * n / dt / 1e12:6.2f} TB/s  last scan kernel {scan:.4f} ms", flush=True)
        if not ok:
            print("   want", want, "got", got[:3])
