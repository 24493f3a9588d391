#!/usr/bin/env python3
"""regex-dna on the GPU: counterpart of the reference's sample/regexdna.cc:25-94 with the text
kept in HBM for the whole program -- one upload, strip (`>.*\\n|\\n` -> ""), nine
MatchAllCount (one fused pass, rj_multi_run), eleven ReplaceAll (IUB codes), three sizes printed --
through the C ABI (rj_scan_run / rj_scan_replace / rj_multi_run).  The reference's own sample also runs unchanged on
librejit_hip.so (oracle/_ref/regexdna_hip), but pays a PCIe round trip per call.

    python samples/regexdna_gpu.py < input.fasta
    python samples/regexdna_gpu.py --n 5000000        # generate the FASTA input on the fly
"""
import argparse
import os
import sys
import time

_T_PROCESS = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=0, help="generate the Benchmarks-Game FASTA input of this size instead of reading stdin")
    ap.add_argument("--timing", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    import rejit_amd
    from rejit_amd import workloads as W
    t_imports = time.perf_counter() - _T_PROCESS

    dev = torch.device("cuda:0")
    t_in = time.perf_counter()
    raw_host = W.fasta_raw_numpy(args.n) if args.n else np.frombuffer(sys.stdin.buffer.read(), dtype=np.uint8)
    t_read = time.perf_counter() - t_in
    t0 = time.perf_counter()
    text = torch.from_numpy(np.ascontiguousarray(raw_host)).to(dev)
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t0    # (the first device call of the process: includes the HIP context)
    stream = torch.cuda.current_stream(dev).cuda_stream
    raw_size = int(text.numel())

    def replace_all(buf, n, regex, repl: bytes):
        sc = rejit_amd.Scan(rejit_amd.Program(regex))
        m = sc.run(buf.data_ptr(), n, stream=stream)
        out = torch.empty(n + m * len(repl) + 64, dtype=torch.uint8, device=dev)
        new_len = sc.replace(buf.data_ptr(), n, repl, out.data_ptr(), int(out.numel()), stream=stream)
        return out, new_len

    phases = []

    def lap(name, since):
        if args.timing:
            torch.cuda.synchronize()
            phases.append("%s %.1f ms" % (name, (time.perf_counter() - since) * 1e3))

    t0 = time.perf_counter()
    text, n = replace_all(text, raw_size, W.REGEXDNA_STRIP, b"")
    text_size = n
    lap("strip", t0)
    t1 = time.perf_counter()
    # the nine counts share one pass over the text (rj_multi, fused window scan)
    multi = rejit_amd.MultiScan([rejit_amd.Program(rx) for rx in W.REGEXDNA_PATTERNS])
    multi.set_counts_only(True)      # MatchAllCount: scan + classification + counts in one kernel
    counts = multi.run(text.data_ptr(), n, stream=stream)
    lines = ["%s %d" % (rx, c) for rx, c in zip(W.REGEXDNA_PATTERNS, counts)]
    lap("9 counts", t1)
    for code, repl in W.REGEXDNA_IUB:
        t2 = time.perf_counter()
        text, n = replace_all(text, n, code, repl.encode())
        lap("replace " + code, t2)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    print("\n".join(lines))
    print("\n%d\n%d\n%d" % (raw_size, text_size, n))
    if args.timing:
        print("; ".join(phases), file=sys.stderr)
        print("imports (numpy, torch, rejit_amd) %.3f s, input %.3f s, HIP context + upload %.3f s, device pipeline (strip + 9 counts + 11 replaces) %.3f s"
              % (t_imports, t_read, t_up, t_dev), file=sys.stderr)


if __name__ == "__main__":
    main()
