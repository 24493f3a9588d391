import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rejit_amd as rj
def text_of(n, alphabet, seed):
    rng = np.random.default_rng(seed)
    return np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), size=n)].copy()
for rx, n, alphabet, breaks in ((b".{0,2}.", 20 << 20, b"abcde", ()), (b".{0,2}.", 66 << 20, b"ab", ()),
                                (b".{0,2}.", 9 << 20, b"abcdefgh", (1 << 20, (1 << 20) + 1, 5 << 20, (8 << 20) + 77)),
                                (b".{0,2}", 6 << 20, b"abc", (4 << 20,)), (b"(a|ab)(c|bcd)?(d*)", 6 << 20, b"abcd", ()),
                                (b"[ab]{1,3}b|.{1,4}c", 5 << 20, b"abcx", (3 << 20,))):
    p = rj.Program(rx)
    t = text_of(n, alphabet, 1234 + n)
    for b in breaks: t[b] = 10
    d = torch.from_numpy(t).cuda()
    s = rj.Scan(p)
    t0 = time.perf_counter(); c = s.run(d.data_ptr(), n); dt = time.perf_counter() - t0
    print(rx, n >> 20, "MiB risk", p.info()["ring_artefact_risk"], "min_len", p.info()["min_len"], "count", c, "%.2f s" % dt, "exact_path", s.stats()["exact_path"], flush=True)
