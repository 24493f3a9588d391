#!/usr/bin/env python3
"""GPU: A/B of the headline step under environment overrides.  usage: step_ab.py [K=V[,K=V...] | -] ...
Each argument is one variant (`-` = no override); every variant runs in a process of its own (the overrides are read
once), the whole list twice so that drift shows.  Prints ms per step of the two-in-flight loop with the tails on the
objects' own streams (bench.py's headline loop), best and median of 5 x 200 steps, and the sum of the nine counts."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    import torch
    import rejit_amd
    from rejit_amd import workloads as W
    dev = torch.device("cuda:0")
    progs = [rejit_amd.Program(p) for p in W.REGEXDNA_PATTERNS]
    main = torch.cuda.current_stream(dev).cuda_stream
    text = W.fasta_stripped_torch(int(os.environ.get("STEP_AB_N", "50000000")), dev)
    n = int(text.numel())
    mode = int(os.environ.get("STEP_AB_MODE", "0"))
    depth = int(os.environ.get("STEP_AB_DEPTH", "2"))
    steps = int(os.environ.get("STEP_AB_STEPS", "200"))
    ms = [rejit_amd.MultiScan(progs) for _ in range(depth)]
    for m in ms:
        m.set_mode(mode)
        m.set_tail_stream(True)
        m.run(text.data_ptr(), n, stream=main)
    torch.cuda.synchronize(dev)
    res = []
    total = 0
    for rep in range(5):
        busy = [False] * depth
        t0 = time.perf_counter()
        for k in range(steps):
            j = k % depth
            if busy[j]:
                total = sum(ms[j].finish())
            ms[j].start(text.data_ptr(), n, stream=main)
            busy[j] = True
        for d in range(depth):
            ms[(steps + d) % depth].finish()
        torch.cuda.synchronize(dev)
        res.append((time.perf_counter() - t0) / steps * 1e3)
    res.sort()
    print("best %.4f median %.4f ms/step  counts %d" % (res[0], res[2], total))
else:
    variants = sys.argv[1:] or ["-"]
    for rnd in range(2):
        for v in variants:
            e = dict(os.environ)
            if v != "-":
                for kv in v.split(","):
                    k, _, val = kv.partition("=")
                    e[k] = val
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True)
            line = out.stdout.strip().split("\n")[-1] if out.stdout.strip() else "FAILED " + out.stderr[-400:]
            print("%-40s %s" % (v, line), flush=True)
