"""A zoo of everyday patterns over 1 GiB of synthetic "log / source" text (words, numbers, punctuation, a line break every ~70 bytes): whole
MatchAll calls, GB/s and the path each took -- where are the cliffs?   python tools/probes/zoo.py [MiB]"""
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
g.manual_seed(5)
# letters are common, digits / punctuation rarer, one line break in ~70 bytes
weights = {c: 30 for c in b"etaoinshrdlucmfwypvbgkqjxz"}
weights.update({c: 6 for c in b"0123456789"})
weights.update({c: 60 for c in b" "})
weights.update({c: 4 for c in b".,:;=-_/@()<>\"'#"})
weights.update({c: 8 for c in b"ETAOINSHR"})
weights[10] = 16
syms = torch.tensor(list(weights.keys()), dtype=torch.uint8, device=dev)
w = torch.tensor([float(v) for v in weights.values()], device=dev)
text = syms[torch.multinomial(w, n, replacement=True, generator=g)].contiguous()
PATTERNS = [b"[a-z]+@[a-z]+", b"[0-9]+\\.[0-9]+", b"[A-Za-z_][A-Za-z0-9_]*", b"//.*", b"#.*", b"[0-9]+-[0-9]+", b"\\([^)]*\\)", b"[a-z]+ing", b" +",
            b"\"[^\"]*\"", b"<[a-z]+>", b"[0-9][0-9]:[0-9][0-9]", b"[a-z]+=[a-z0-9]+", b"error", b"(error|warning|fatal)", b"[A-Z][a-z]+", b"[a-z]+\\.[a-z]+",
            b"^[a-z]+", b"[a-z]+$", b"the [a-z]+", b"[0-9]+", b"0x[0-9a-f]+", b"[a-z]+[0-9]+[a-z]+", b".*error.*", b"[^ ]+@[^ ]+",
            b"a.*b", b"<[^>]*>", b"a.+b", b"<[^>]+>", b"#.+", b"@[a-z]+", b"[a-z][a-z0-9]+", b"^#.*", b"#.*$", b"^a.*b", b"[A-Z][a-z]+$", b"^[A-Z][a-z]+", b" +$", b"^ +", b"^[a-z]+$"]
if len(sys.argv) > 2:
    PATTERNS = [p.encode() for p in sys.argv[2:]]
for rx in PATTERNS:
    try:
        sc = rejit_amd.Scan(rejit_amd.Program(rx))
        t0 = time.perf_counter()
        k = sc.run_tensor(text)
        first = time.perf_counter() - t0
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            k = sc.run_tensor(text)
            ts.append(time.perf_counter() - t0)
        st = sc.stats()
        path = "run" if st["run_path"] == 1 else "pair" if st["run_path"] == 2 else "stream" if st["stream_path"] else "linear" if st["linear_path"] else "exact" if st["exact_path"] else "general"
        print(f"{rx.decode():28s} {k:>10d} matches  first {first * 1e3:9.3f} ms  best {min(ts) * 1e3:9.3f} ms = {n / min(ts) / 1e9:8.1f} GB/s  {path}  retries {st['retries']} slow {st['slow_starts']} kernels {st['scan_ms']:.3f} ms", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{rx.decode():28s} ERROR {e!r}"[:200], flush=True)
