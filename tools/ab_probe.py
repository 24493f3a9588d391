#!/usr/bin/env python3
"""A/B of scan-kernel launch geometry (env RJ_SCAN_GRID = workgroups)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, rejit_amd
    from rejit_amd import workloads as W
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    def run(rx, t, n, tag):
        sc = rejit_amd.Scan(rejit_amd.Program(rx))
        for _ in range(3): sc.run(t.data_ptr(), n, stream=st)
        ms = []
        for _ in range(7):
            sc.run(t.data_ptr(), n, stream=st); ms.append(sc.stats()["scan_ms"])
        res[tag] = round(n / min(ms) / 1e6)
    n = 5_000_000_000
    t = W.random_ascii_torch(n, 1, dev)
    run("rege", t, n, "rege5G"); run("regexp", t, n, "regexp5G"); run("regexp", t, n // 10, "regexp.5G")
    del t
    f = W.fasta_stripped_torch(50_000_000, dev)
    run(W.REGEXDNA_PATTERNS[0], f, f.numel(), "dna1"); run(W.REGEXDNA_PATTERNS[4], f, f.numel(), "dna5")
    print(json.dumps(res))
else:
    variants = [{}] + [{"RJ_SCAN_GRID": str(g)} for g in (2048, 4096, 8192, 16384, 32768)]
    for env in variants:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        print(env, out.stdout.strip().split("\n")[-1] if out.stdout.strip() else out.stderr[-300:])
