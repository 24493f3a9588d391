"""Timing probe of the linear-time carry scan (rocprofv3 --kernel-trace --stats around it gives the
per-kernel split): pattern, text size -> wall ms, stats."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rejit_amd

dev = torch.device("cuda:0")
cases = [(b"[acgt]+", b"acgt", 64 << 20), (b"a.*b", b"abcdefgh", 64 << 20), (b"x*", b"xy", 16 << 20),
         (b"[ab]{40}c*", b"ab", 4 << 20), (b"[ab]{70,90}b*", b"ab", 1 << 20)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0].decode() in sys.argv[1:]]
for rx, alphabet, n in cases:
    g = torch.Generator(device="cuda").manual_seed(1)
    lut = torch.tensor(list(alphabet), dtype=torch.uint8, device=dev)
    d = lut[torch.randint(0, len(alphabet), (n,), generator=g, device=dev)].contiguous()
    sc = rejit_amd.Scan(rejit_amd.Program(rx))
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c = sc.run_tensor(d)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = sc.stats()
        print(rx.decode(), n >> 20, "MiB run", it, "matches", c, "wall %.1f ms" % (dt * 1e3), "scan_ms %.1f" % st["scan_ms"],
              "linear", st["linear_path"], "retries", st["retries"], "GB/s %.2f" % (n / dt / 1e9), flush=True)
