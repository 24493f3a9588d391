"""Whole calls of `"[^"]*"` / `"[^"\\n]*"` over a JSON-like text (a quote every ~32 bytes, a line break every ~64) and over a text of a few
long strings: the pair kernels (run_scan.hip) against the paths the patterns took before (RJ_NO_PAIRS=1: run this script twice).
    python tools/probes/pair_time.py [MiB]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import rejit_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda")
g.manual_seed(3)
alphabet = torch.tensor(list((b"\"\"\n" + b"abcdefghijklmnopqrstuvwxyz0123456789 ,:{}[]_-.ABCDEFGHIJKLMNOPQRS")[:64]), device=dev, dtype=torch.uint8)
json_like = alphabet[torch.randint(0, 64, (n,), device=dev, generator=g).long()].contiguous()
sparse = torch.full((n,), ord("x"), dtype=torch.uint8, device=dev)
sparse[torch.randint(0, n, (n // 30000,), device=dev, generator=g)] = ord("\"")
for name, d in (("json-like", json_like), ("long strings", sparse)):
    for rx in (b"\"[^\"]*\"", b"\"[^\"\n]*\""):
        sc = rejit_amd.Scan(rejit_amd.Program(rx))
        k = sc.run_tensor(d)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = sc.run_tensor(d)
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        kc = sc.count_tensor(d)
        tc = time.perf_counter() - t0
        st = sc.stats()
        print(f"{name:13s} {rx.decode().replace(chr(10), '<LF>'):14s} {mib} MiB: {k} matches, best call {min(ts) * 1e3:8.3f} ms = {n / min(ts) / 1e9:7.1f} GB/s; count {kc} in {tc * 1e3:8.3f} ms; "
              f"run_path {st['run_path']} linear {st['linear_path']} stream {st['stream_path']}", flush=True)
