"""GPU parity tests (-m gpu) of the PAIR kernels (rejit_amd/csrc/run_scan.h / run_scan.hip: pair_summary, pair_resolve, pair_emit): `Q L* Q`
with the same class at both ends and no Q inside L -- `"[^"]*"`, `'[^'\\n]*'`, `%[a-z]*%` -- whose matches are the pairs of the Q bytes
since the last reset (reference: the thread a Q opens lives until the next break, src/x64/codegen-x64.cc:535-581; the match ends behind a
closing Q, :426-461, and the scan restarts behind it, :487-503), against the oracle through the C ABI:

* texts of every density of Q bytes and resets, sizes around the kernels' units (32 B words, 2 KiB iterations, 8 KiB tiles), Q bytes and
  resets packed into single words, strings that cross iterations, tiles and resolve chunks, an unmatched Q at the text's end;
* a 48 MiB JSON-like text (32 KiB tiles, the two-level resolve) by properties and against the oracle, a 64 MiB text that is ONE string;
* MatchAllCount (rj_scan_count) stops behind the resolve; own ranges keep the other paths and still agree.
"""
import random

import numpy as np
import pytest

from checkers import Oracle

pytestmark = pytest.mark.gpu

# (the reference's dialect has no escapes inside brackets: a line break in a class is the byte itself)
SHAPES = [b"\"[^\"]*\"", b"'[^'\n]*'", b"%[a-z]*%", b"[\"'][^\"']*[\"']", b"\\|[^|\r\n]*\\|", b"[\x80][^\x80]*[\x80]", b"\"[^\"\n]*\""]


@pytest.fixture(scope="module")
def rj():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import rejit_amd
    rejit_amd.build()
    rejit_amd.load_library()
    return rejit_amd


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def device_text(data: bytes):
    import torch
    return torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()


def texts_for(rng, n):
    out = []
    for alphabet in (b"ab\"", b"\"'", b"ab\"'%|\n ", b"abcdefgh\"\n", b"\"\n", b"a%b%c \x80", bytes(range(256)), b"abcxyz"):
        out.append(bytes(rng.choice(alphabet) for _ in range(n)))
    # long strings: a Q byte / a reset every few thousand bytes
    for every in (700, 5000, 40000):
        t = bytearray(rng.choice(b"abcxyz ") for _ in range(n))
        for p in range(rng.randrange(every), n, every):
            t[p] = rng.choice(b"\"\"\"'%|\n\x80")
        out.append(bytes(t))
    return out


def check(rj, oracle, rx, data, want_path=None, **kw):
    sc = rj.Scan(rj.Program(rx))
    t = device_text(data)
    n = len(data)
    got_n = sc.run(t.data_ptr(), n, **kw) if n else 0
    got = sc.spans() if n else []
    want = oracle.match_all(rx, data)
    if "own_begin" in kw:
        ob, oe = kw["own_begin"], kw.get("own_end", n + 1)
        alt = [(b + ob, e + ob) for b, e in oracle.match_all(rx, data[ob:]) if b + ob < oe]
        want = alt if got == alt else [m for m in want if ob <= m[0] < oe]
    assert got == want and got_n == len(want), (rx, n, kw, got[:4], want[:4], len(got), len(want))
    st = sc.stats()
    if want_path is not None:
        assert st["run_path"] == want_path, (rx, n, st)
    return st


def test_pair_shapes_vs_oracle(rj, oracle):
    rng = random.Random(61)
    took = 0
    for n in (17000, 20479, 20480, 20481, 24576, 65536, 70001, 300000):
        for data in texts_for(rng, n):
            for rx in SHAPES:
                took += 1 if check(rj, oracle, rx, data)["run_path"] == 2 else 0
    assert took > 300, took


def test_packed_words_and_edges(rj, oracle):
    """Q bytes and resets packed into single 32-byte words, at word / iteration / tile edges, an open Q at the end of the text."""
    rng = random.Random(62)
    n = 40000
    for rx in (b"\"[^\"\n]*\"", b"\"[^\"]*\""):
        for trial in range(12):
            t = bytearray(b"a" * n)
            for edge in (2048, 4096, 8192, 16384, 24576, 32768):
                for d in range(-34, 35):
                    if rng.random() < (0.15, 0.5, 0.9)[trial % 3]:
                        t[edge + d] = rng.choice(b"\"\"\n")
            for _ in range(trial * 3):
                p = rng.randrange(n - 40)
                for d in range(rng.randint(1, 33)):
                    t[p + d] = rng.choice(b"\"\"\"\na")
            if trial % 2:
                t[n - 1 - rng.randrange(3)] = ord("\"")
            check(rj, oracle, rx, bytes(t), want_path=2)
    # every byte a Q; every byte a reset; one Q; none
    check(rj, oracle, b"\"[^\"]*\"", b"\"" * 33333, want_path=2)
    check(rj, oracle, b"\"[^\"\n]*\"", b"\n" * 33333, want_path=2)
    check(rj, oracle, b"\"[^\"]*\"", b"a" * 20000 + b"\"" + b"a" * 20000, want_path=2)
    check(rj, oracle, b"\"[^\"]*\"", b"a" * 40000, want_path=2)


def test_counts_and_ranges(rj, oracle):
    rng = random.Random(63)
    data = bytes(rng.choice(b"abcdefgh\"\"\n ,:{}") for _ in range(500000))
    for rx in (b"\"[^\"]*\"", b"\"[^\"\n]*\""):
        want = oracle.match_all(rx, data)
        sc = rj.Scan(rj.Program(rx))
        t = device_text(data)
        assert sc.count(t.data_ptr(), len(data)) == len(want)
        assert sc.stats()["run_path"] == 2
        # own ranges (the sharded shape) keep the other paths and agree
        for ob, oe in ((0, 250000), (123457, 400001), (250000, len(data) + 1)):
            check(rj, oracle, rx, data, own_begin=ob, own_end=oe)


def test_json_like_48mib_and_one_long_string(rj, oracle):
    """32 KiB tiles and the two-level resolve; properties first (cheap), then the oracle."""
    import torch
    n = 48 << 20
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    r = torch.randint(0, 64, (n,), device="cuda", generator=g, dtype=torch.int32)
    alphabet = torch.tensor(list((b"\"\"\n" + b"abcdefghijklmnopqrstuvwxyz0123456789 ,:{}[]_-.ABCDEFGHIJKLMNOPQRS")[:64]), device="cuda", dtype=torch.uint8)
    d = alphabet[r.long()].contiguous()
    for rx in (b"\"[^\"]*\"", b"\"[^\"\n]*\""):
        sc = rj.Scan(rj.Program(rx))
        k = sc.run_tensor(d)
        sp = sc.spans()
        assert sc.stats()["run_path"] == 2 and k == len(sp)
        h = d.cpu().numpy().tobytes()
        q = np.frombuffer(h, dtype=np.uint8) == ord("\"")
        arr = np.array(sp, dtype=np.int64).reshape(-1, 2)
        # every match begins and ends on a quote, holds no other, and the matches are disjoint and ordered
        assert q[arr[:, 0]].all() and q[arr[:, 1] - 1].all() and (arr[1:, 0] >= arr[:-1, 1]).all()
        cs = np.concatenate([[0], np.cumsum(q)])
        assert ((cs[arr[:, 1]] - cs[arr[:, 0]]) == 2).all()
        if rx == b"\"[^\"]*\"":
            assert k == int(q.sum()) // 2
        assert sp == oracle.match_all(rx, h)
    # ONE string of 64 MiB, and the same text with its closing quote gone
    n = 64 << 20
    d = torch.full((n,), ord("x"), dtype=torch.uint8, device="cuda")
    d[5] = ord("\"")
    d[n - 7] = ord("\"")
    sc = rj.Scan(rj.Program(b"\"[^\"]*\""))
    assert sc.run_tensor(d) == 1 and sc.spans() == [(5, n - 6)] and sc.stats()["run_path"] == 2
    d[n - 7] = ord("x")
    assert sc.run_tensor(d) == 0 and sc.spans() == []


def test_beyond_4gib(rj):
    """64-bit positions: strings that begin, end and span the 4 GiB line in a 4.5 GiB text (131 072+ tiles: the two-level resolve)."""
    import torch
    n = (9 << 29) + 12345
    d = torch.full((n,), ord("x"), dtype=torch.uint8, device="cuda")
    four = 1 << 32
    quotes = [5, 100, 70000, four - 3, four + 5, four + 6, n - 10, n - 2]
    for p in quotes:
        d[p] = ord("\"")
    want = [(quotes[i], quotes[i + 1] + 1) for i in range(0, len(quotes), 2)]
    sc = rj.Scan(rj.Program(b"\"[^\"]*\""))
    assert sc.run_tensor(d) == len(want) and sc.spans() == want and sc.stats()["run_path"] == 2
    assert sc.count_tensor(d) == len(want)
    # a line break inside the second string: with `\n` a reset its opening quote is dropped and the quotes behind pair up afresh --
    # (four - 3, four + 5) spans the 4 GiB line, (four + 6, n - 10) the last 0.5 GiB
    sc2 = rj.Scan(rj.Program(b"\"[^\"\n]*\""))
    d[80000] = ord("\n")
    want2 = [(5, 101), (four - 3, four + 6), (four + 6, n - 9)]
    assert sc2.run_tensor(d) == len(want2) and sc2.spans() == want2
    # one quote more: it stays open at the text's end
    d[n - 1] = ord("\"")
    assert sc.run_tensor(d) == 4 and sc.spans() == want
