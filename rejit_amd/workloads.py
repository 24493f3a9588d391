"""Synthetic inputs for the BASELINE.json configurations.

Everything here is a pure function of (size, seed), with a numpy implementation for
CPU-sized inputs and a torch implementation that builds the same bytes directly in HBM
for GPU-sized ones (there is no network, and a 5-50 GB text does not come over PCIe at
HBM speed).  torch is used only as the device allocator / elementwise engine.

* `fasta_*`       -- the public Benchmarks-Game `fasta` generator (NOT part of the
                     reference; sample/regexdna.cc:35-36 only links to it), restated
                     from its published algorithm: LCG IM=139968 IA=3877 IC=29573 seed 42,
                     ALU repeat x2n, IUB x3n, homo-sapiens x5n, 60-column lines.  The LCG
                     has full period 139968 (Hull-Dobell), so character k of a random
                     section is a table lookup at (k + offset) mod 139968 -- which is what
                     makes a data-parallel device generator possible.
* `random_ascii`  -- i.i.d. bytes uniform over [lo, hi) from a counter-based mixer
                     (splitmix64 finaliser of the byte index), the distribution of the
                     reference's benchmark harness (tools/benchmarks/engines/
                     bench_engine.cc:201-207: low + rand() % (high - low), default range
                     ['0','z'), tools/benchmarks/run.py:313).
* `plant`         -- overwrite the text with known occurrences at known offsets.
"""
from __future__ import annotations

import random as _random
from typing import List, Sequence, Tuple

import numpy as np

IM, IA, IC = 139968, 3877, 29573
LCG_SEED = 42
LINE = 60

ALU = (
    "GGCCGGGCGCGGTGGCTCACGCCTGTAATCCCAGCACTTTGG"
    "GAGGCCGAGGCGGGCGGATCACCTGAGGTCAGGAGTTCGAGA"
    "CCAGCCTGGCCAACATGGTGAAACCCCGTCTCTACTAAAAAT"
    "ACAAAAATTAGCCGGGCGTGGTGGCGCGCGCCTGTAATCCCA"
    "GCTACTCGGGAGGCTGAGGCAGGAGAATCGCTTGAACCCGGG"
    "AGGCGGAGGTTGCAGTGAGCCGAGATCGCGCCACTGCACTCC"
    "AGCCTGGGCGACAGAGCGAGACTCCGTCTCAAAAA"
)
IUB = [("a", 0.27), ("c", 0.12), ("g", 0.12), ("t", 0.27)] + [(c, 0.02) for c in "BDHKMNRSVWY"]
HS = [("a", 0.3029549426680), ("c", 0.1979883004921), ("g", 0.1975473066391), ("t", 0.3015094502008)]

HEADERS = [b">ONE Homo sapiens alu\n", b">TWO IUB ambiguity codes\n", b">THREE Homo sapiens frequency\n"]

REGEXDNA_PATTERNS = [
    "agggtaaa|tttaccct",
    "[cgt]gggtaaa|tttaccc[acg]",
    "a[act]ggtaaa|tttacc[agt]t",
    "ag[act]gtaaa|tttac[agt]ct",
    "agg[act]taaa|ttta[agt]cct",
    "aggg[acg]aaa|ttt[cgt]ccct",
    "agggt[cgt]aa|tt[acg]accct",
    "agggta[cgt]a|t[acg]taccct",
    "agggtaa[cgt]|[acg]ttaccct",
]
REGEXDNA_STRIP = ">.*\n|\n"
REGEXDNA_IUB = [("B", "(c|g|t)"), ("D", "(a|g|t)"), ("H", "(a|c|t)"), ("K", "(g|t)"), ("M", "(a|c)"),
                ("N", "(a|c|g|t)"), ("R", "(a|g)"), ("S", "(c|g)"), ("V", "(a|c|g)"), ("W", "(a|t)"),
                ("Y", "(c|t)")]


def _lcg_orbit() -> np.ndarray:
    """orbit[k] = LCG state after k steps from seed 42; orbit has full period IM."""
    orbit = np.empty(IM, dtype=np.int64)
    s = LCG_SEED
    for k in range(IM):
        orbit[k] = s
        s = (s * IA + IC) % IM
    assert s == LCG_SEED, "LCG is expected to have full period"
    return orbit


_ORBIT = None
_TABLES = {}


def _char_table(freqs) -> np.ndarray:
    """table[k] = character selected by the k-th LCG draw (k >= 1 is the first draw)."""
    global _ORBIT
    if _ORBIT is None:
        _ORBIT = _lcg_orbit()
    key = tuple(freqs)
    if key not in _TABLES:
        cum, acc = [], 0.0
        for _, p in freqs:
            acc += p
            cum.append(acc)
        r = _ORBIT.astype(np.float64) / float(IM)
        idx = np.searchsorted(np.array(cum), r, side="right")  # first cum strictly > r
        idx = np.minimum(idx, len(freqs) - 1)
        chars = np.frombuffer("".join(c for c, _ in freqs).encode(), dtype=np.uint8)
        _TABLES[key] = chars[idx]
    return _TABLES[key]


def fasta_stripped_size(n: int) -> int:
    return 10 * n


def fasta_stripped_numpy(n: int) -> np.ndarray:
    """The FASTA text after regexdna's strip step (headers and newlines removed):
    2n ALU + 3n IUB + 5n HS characters."""
    alu = np.frombuffer(ALU.encode(), dtype=np.uint8)
    k = np.arange(2 * n, dtype=np.int64)
    one = alu[k % len(alu)]
    iub, hs = _char_table(IUB), _char_table(HS)
    j = np.arange(3 * n, dtype=np.int64)
    two = iub[(j + 1) % IM]
    j = np.arange(5 * n, dtype=np.int64)
    three = hs[(j + 1 + 3 * n) % IM]
    return np.concatenate([one, two, three])


def _with_lines(seq: np.ndarray) -> np.ndarray:
    n = len(seq)
    nl = (n + LINE - 1) // LINE
    out = np.full(n + nl, ord("\n"), dtype=np.uint8)
    k = np.arange(n, dtype=np.int64)
    out[k + k // LINE] = seq
    return out


def fasta_raw_numpy(n: int) -> np.ndarray:
    """The FASTA file exactly as the Benchmarks-Game program prints it."""
    s = fasta_stripped_numpy(n)
    parts = []
    bounds = [0, 2 * n, 5 * n, 10 * n]
    for i in range(3):
        parts.append(np.frombuffer(HEADERS[i], dtype=np.uint8))
        parts.append(_with_lines(s[bounds[i]:bounds[i + 1]]))
    return np.concatenate(parts)


def fasta_stripped_torch(n: int, device, chunk: int = 1 << 28, lo: int = 0, hi: int = None):
    """Bytes [lo, hi) of fasta_stripped_numpy(n), built on `device` (uint8 tensor).  A rank
    of a sharded run generates only its own range (+ halo) this way."""
    import torch

    total = 10 * n
    hi = total if hi is None else hi
    lo, hi = max(0, lo), min(total, hi)
    out = torch.empty(max(hi - lo, 0), dtype=torch.uint8, device=device)
    alu = torch.from_numpy(np.frombuffer(ALU.encode(), dtype=np.uint8).copy()).to(device)
    iub = torch.from_numpy(_char_table(IUB).copy()).to(device)
    hs = torch.from_numpy(_char_table(HS).copy()).to(device)
    sections = [(0, 2 * n, alu, 0, len(ALU)), (2 * n, 3 * n, iub, 1, IM), (5 * n, 5 * n, hs, 1 + 3 * n, IM)]
    for base, count, table, off, mod in sections:
        a, b = max(lo, base), min(hi, base + count)   # overlap of [lo,hi) with this section
        for c0 in range(a, b, chunk):
            c1 = min(b, c0 + chunk)
            k = torch.arange(c0 - base, c1 - base, dtype=torch.int64, device=device)
            out[c0 - lo:c1 - lo] = table[(k + off) % mod]
    return out


# ---------------------------------------------------------------------------
# Random ASCII

def fasta_raw_torch(n: int, device):
    """fasta_raw_numpy(n) built on `device`: the three headers and a line break after every 60 bases."""
    import torch

    s = fasta_stripped_torch(n, device)
    parts = []
    bounds = [0, 2 * n, 5 * n, 10 * n]
    for i in range(3):
        seq = s[bounds[i]:bounds[i + 1]]
        m = int(seq.numel())
        out = torch.full((m + (m + LINE - 1) // LINE,), ord("\n"), dtype=torch.uint8, device=device)
        k = torch.arange(m, dtype=torch.int64, device=device)
        out[k + k // LINE] = seq
        del k
        parts.append(torch.tensor(list(HEADERS[i]), dtype=torch.uint8, device=device))
        parts.append(out)
    return torch.cat(parts)


_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_GOLD = 0x9E3779B97F4A7C15
_MASK = (1 << 64) - 1


def _mix_numpy(idx: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = idx.astype(np.uint64) + np.uint64((seed * _GOLD + _GOLD) & _MASK)
        x ^= x >> np.uint64(30)
        x *= np.uint64(_M1)
        x ^= x >> np.uint64(27)
        x *= np.uint64(_M2)
        x ^= x >> np.uint64(31)
    return x


def random_ascii_numpy(n: int, seed: int, lo: int = ord("0"), hi: int = ord("z"), start: int = 0) -> np.ndarray:
    """bytes[start : start+n] of the infinite stream for `seed`, uniform over [lo, hi)."""
    x = _mix_numpy(np.arange(start, start + n, dtype=np.uint64), seed)
    return (np.uint64(lo) + (x >> np.uint64(33)) % np.uint64(hi - lo)).astype(np.uint8)


def _signed(v: int) -> int:
    v &= _MASK
    return v - (1 << 64) if v >= (1 << 63) else v


def random_ascii_torch(n: int, seed: int, device, lo: int = ord("0"), hi: int = ord("z"), start: int = 0,
                       chunk: int = 1 << 28):
    """Same stream as random_ascii_numpy, generated on `device` in chunks (int64
    arithmetic wraps like uint64; logical shifts are emulated with masks)."""
    import torch

    out = torch.empty(n, dtype=torch.uint8, device=device)
    add = _signed(seed * _GOLD + _GOLD)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        x = torch.arange(start + a, start + b, dtype=torch.int64, device=device) + add
        x = x ^ ((x >> 30) & ((1 << 34) - 1))
        x = x * _signed(_M1)
        x = x ^ ((x >> 27) & ((1 << 37) - 1))
        x = x * _signed(_M2)
        x = x ^ ((x >> 31) & ((1 << 33) - 1))
        u = (x >> 33) & ((1 << 31) - 1)
        out[a:b] = (lo + u % (hi - lo)).to(torch.uint8)
        del x, u
    return out


def plant_offsets(n: int, length: int, count: int, seed: int, boundaries: Sequence[int] = ()) -> List[int]:
    """`count` sorted, pairwise non-overlapping, non-adjacent offsets for planted
    occurrences: offset 0, the very end, a back-to-back pair, one straddling each
    requested boundary (16 B / 1 KiB / tile / shard edges ...), and random ones."""
    rng = _random.Random(seed)
    want = [0, n - length]
    for b in boundaries:
        for d in (1, length // 2, length - 1):
            if 0 <= b - d and b - d + length <= n:
                want.append(b - d)
    picked: List[int] = []

    def ok(o):
        return 0 <= o <= n - length and all(abs(o - p) > length for p in picked)

    for o in want:
        if ok(o):
            picked.append(o)
    # one adjacent pair (back to back, distance == length)
    for _ in range(100):
        o = rng.randrange(0, max(1, n - 2 * length))
        if ok(o) and ok(o + length) and len(picked) + 2 <= count:
            picked.append(o)
            picked.append(o + length)
            break
    tries = 0
    while len(picked) < count and tries < 100 * count:
        tries += 1
        o = rng.randrange(0, n - length + 1)
        if ok(o):
            picked.append(o)
    return sorted(picked)[:max(count, 0)] if len(picked) > count else sorted(picked)


def plant(text, offsets: Sequence[int], needle: bytes):
    """In-place: write `needle` at every offset (numpy array or torch tensor)."""
    if isinstance(text, np.ndarray):
        nd = np.frombuffer(needle, dtype=np.uint8)
        for o in offsets:
            text[o:o + len(needle)] = nd
    else:
        import torch

        nd = torch.tensor(list(needle), dtype=torch.uint8, device=text.device)
        idx = torch.tensor(list(offsets), dtype=torch.int64, device=text.device)
        pos = (idx[:, None] + torch.arange(len(needle), device=text.device)[None, :]).reshape(-1)
        text[pos] = nd.repeat(len(offsets))
    return text


# ---- a digest of a span list that numpy (the reference's output, tests/golden/make_fullsize.py) and torch (the
# engine's output, left on the device) compute alike: 64-bit wrapping sums.  Pins results at BASELINE sizes, where
# the span lists themselves are too large for a fixture.
_D1 = 0x9E3779B97F4A7C15
_D2 = 0xC2B2AE3D27D4EB4F


def span_digest_numpy(spans: np.ndarray) -> dict:
    """spans: (k, 2) uint64."""
    b, e = spans[:, 0].astype(np.uint64), spans[:, 1].astype(np.uint64)
    with np.errstate(over="ignore"):
        mix = ((b * np.uint64(_D1)) ^ (e * np.uint64(_D2))).sum(dtype=np.uint64)
        return {"count": int(len(b)), "sum_begin": int(b.sum(dtype=np.uint64)), "sum_len": int((e - b).sum(dtype=np.uint64)),
                "mix": int(mix)}


def span_digest_torch(spans) -> dict:
    """spans: (k, 2) int64 tensor (any device); int64 arithmetic wraps like uint64."""
    b, e = spans[:, 0], spans[:, 1]
    mix = ((b * _signed(_D1)) ^ (e * _signed(_D2))).sum()
    return {"count": int(spans.shape[0]), "sum_begin": int(b.sum()) & _MASK, "sum_len": int((e - b).sum()) & _MASK, "mix": int(mix) & _MASK}


# Strings drawn from the language of the "complex" benchmark regex
# ([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)   (tools/benchmarks/run.py:351)
def complex_regex_sample(rng: _random.Random) -> bytes:
    reps = rng.randint(2, 7)
    parts = []
    for _ in range(reps):
        parts.append(b"regexp" if rng.random() < 0.5 else bytes([rng.choice(b"complex")]))
    tail = rng.choice([b"at", b"the", bytes([rng.choice(b"efghijklmnd")]) + b"as well"])
    return b"".join(parts) + b"abcdefgh" + tail


BENCH_REGEXES: List[Tuple[str, str, str]] = [
    # (regexp, low_char, high_char)  tools/benchmarks/run.py:347-360
    ("abcdefgh", "b", "z"),
    ("abcdefgh", "0", "z"),
    ("abcdefgh", "a", "j"),
    ("([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)", "0", "z"),
    ("(alternation|strings)", "0", "z"),
    ("(alternation|more|than|two|different|strings)", "0", "z"),
    ("(rather_long_string|min)", "0", "z"),
    ("(([complex]|(regexp)){2,7}alternation)|(strings(at|the|[e-nd]as well))", "0", "z"),
    ("(prefix abcd|prefix 1234)", "0", "z"),
    ("(abcd suffix|1234 suffix)", "0", "z"),
    ("(abcdefgh anywhere xyz|01 anywhere 56789)", "0", "z"),
    ("(some|[stuff])((other|regexps)? bla root blah | (abcdefgh boot{3,3} xyz | 00 foot 5678))", "0", "z"),
]
