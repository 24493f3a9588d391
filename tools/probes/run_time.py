import sys, time; sys.path.insert(0, "/root/repo")
import torch, rejit_amd
dev = torch.device("cuda:0")
for rx, alphabet, n in [(b"[acgt]+", b"acgt", 64 << 20), (b"a.*b", b"abcdefgh", 64 << 20), (b"[acgt]+", b"acgt", 4 << 30), (b"[acgt]+", b"acgtacgtacgtacgtacgtacgtacgtacgtN", 4 << 30)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    lut = torch.tensor(list(alphabet), dtype=torch.uint8, device=dev)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for lo in range(0, n, 1 << 28):
        hi = min(n, lo + (1 << 28))
        d[lo:hi] = lut[torch.randint(0, len(alphabet), (hi - lo,), generator=g, device=dev)]
    sc = rejit_amd.Scan(rejit_amd.Program(rx))
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); c = sc.run_tensor(d); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = sc.stats()
    print(rx.decode(), len(alphabet), "letters", n >> 20, "MiB: matches", c, "wall %.3f ms = %.0f GB/s" % (dt * 1e3, n / dt / 1e9), "run_path", st["run_path"], "stream", st["stream_path"], flush=True)
    del d
