import ctypes, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, rejit_amd
from rejit_amd import workloads as W
from rejit_amd.api import load_library
n = 1_000_000_000
t = W.random_ascii_torch(n, 3, torch.device("cuda:0")).cpu().numpy()
W.plant(t, W.plant_offsets(n, 6, 1000, seed=3), b"regexp")
tb = t.tobytes(); del t
L = load_library()
q = rejit_amd.Program(b"regexp")
for rep in range(3):
    t0 = time.perf_counter(); c = q.count(tb); t1 = time.perf_counter()
    out = ctypes.c_void_p(); out_len = ctypes.c_size_t()
    m = L.rj_replace_all(q._h, tb, len(tb), b"REGEXP!", 7, ctypes.byref(out), ctypes.byref(out_len))
    t2 = time.perf_counter()
    L.rj_free_text(out)
    t3 = time.perf_counter()
    print("count %.1f ms; rj_replace_all %.1f ms (m=%d, out_len=%d); free %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, m, out_len.value, (t3-t2)*1e3), flush=True)
# bare D2H into fresh malloc
libc = ctypes.CDLL("libc.so.6"); libc.malloc.restype = ctypes.c_void_p; libc.malloc.argtypes=[ctypes.c_size_t]; libc.free.argtypes=[ctypes.c_void_p]
hip = ctypes.CDLL("libamdhip64.so.7"); hip.hipMemcpy.argtypes=[ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
for rep in range(3):
    h = libc.malloc(n)
    t0 = time.perf_counter(); hip.hipMemcpy(h, d.data_ptr(), n, 2); t1 = time.perf_counter()
    hip.hipMemcpy(h, d.data_ptr(), n, 2); t2 = time.perf_counter()
    libc.free(h)
    print("bare D2H into fresh malloc: %.1f ms; again into the same (touched) pages: %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3), flush=True)
