"""The run kernels over texts with a break every ~k bytes (`[acgt]+` over acgt with an N now and then; `a.*b` over lines of k bytes):
wall time of a whole call.  usage: run_density.py [MiB]"""
import os, sys, time; sys.path.insert(0, "/root/repo")
os.environ["RJ_RUNS_FIRST"] = "1"
import torch, rejit_amd
dev = torch.device("cuda:0")
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
g = torch.Generator(device="cuda").manual_seed(1)
for rx, alphabet, brk in [(b"[acgt]+", b"acgt", ord("N")), (b"a.*b", b"abcdefgh", 10)]:
    lut = torch.tensor(list(alphabet), dtype=torch.uint8, device=dev)
    for k in (33, 80, 300, 1000, 4000, 30000):
        d = lut[torch.randint(0, len(alphabet), (n,), generator=g, device=dev)]
        d[torch.rand(n, generator=g, device=dev) < 1.0 / k] = brk
        sc = rejit_amd.Scan(rejit_amd.Program(rx))
        best = 1e9
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); c = sc.run_tensor(d); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        st = sc.stats()
        print("%-8s break every ~%-6d %5d MiB: matches %10d  %.3f ms = %5.0f GB/s  run_path %d" % (rx.decode(), k, n >> 20, c, best * 1e3, n / best / 1e9, st["run_path"]), flush=True)
        del d
