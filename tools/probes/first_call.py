"""The FIRST call of a fresh scan object against the later ones (allocation, list growth, retries): python tools/probes/first_call.py [MiB] pattern..."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import rejit_amd
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "zoo.py")).read()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
exec(src.split("PATTERNS = [")[0].replace("mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024", "pass"))
for rx in [p.encode() for p in sys.argv[2:]]:
    for trial in range(2):
        sc = rejit_amd.Scan(rejit_amd.Program(rx))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = sc.run_tensor(text)
        dt = time.perf_counter() - t0
        st = sc.stats()
        t0 = time.perf_counter()
        k = sc.run_tensor(text)
        dt2 = time.perf_counter() - t0
        print(f"{rx.decode():24s} trial {trial}: first {dt * 1e3:9.3f} ms (retries {st['retries']} large {st['large_path']} kernels {st['scan_ms']:.3f} total {st['total_ms']:.3f}) second {dt2 * 1e3:9.3f} ms, {k} matches", flush=True)
