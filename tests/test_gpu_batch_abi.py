"""GPU tests (-m gpu) of the batch / packed / asynchronous entry points of include/rejit_hip.h that round 3 added without
a driver-run test: rj_match_all_packed + rj_batch_separator + rj_host_alloc / rj_host_free (a batch the caller lays out
itself), rj_scan_start / rj_scan_finish with two runs in flight, and the native grep samples/jrep_gpu (C++ over the C ABI)
against GNU grep and the real reference's jrep (oracle/_ref/jrep_ref, built in the build container; it travels).

Reference behaviour these replace: one MatchAll per file, sample/jrep.cc:261-313; output format :336-369."""
import ctypes
import os
import random
import shutil
import subprocess

import pytest

from checkers import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


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


def _layout(texts, sep, lead, gaps, rng):
    """texts -> (bytes of the packed buffer, offsets, sizes): `lead` separator bytes before the first text, 1..gaps behind
    every text (at least one: the contract of rj_match_all_packed)."""
    buf = bytearray(bytes([sep]) * lead)
    offsets, sizes = [], []
    for t in texts:
        offsets.append(len(buf))
        sizes.append(len(t))
        buf += t
        buf += bytes([sep]) * rng.randint(1, gaps)
    return bytes(buf), offsets, sizes


def _packed(rj, p, buf, offsets, sizes, pinned):
    lib = rj.load_library()
    if pinned:
        ptr = lib.rj_host_alloc(len(buf) + 64)
        assert ptr, "rj_host_alloc returned NULL on a GPU box"
        ctypes.memmove(ptr, buf, len(buf))
        try:
            return p.match_all_packed(ptr, offsets, sizes, len(buf))
        finally:
            lib.rj_host_free(ptr)
    keep = ctypes.create_string_buffer(buf, len(buf))
    return p.match_all_packed(ctypes.addressof(keep), offsets, sizes, len(buf))


def test_packed_batches_equal_per_text_oracle(rj, oracle):
    """rj_match_all_packed == the oracle text by text: literal, class, dense, nullable (`x*`, `^`, `$`: empty matches inside
    the gaps belong to no text) and a pattern without a usable separator; a leading gap, gaps of several bytes,
    texts of 0 bytes, pinned (rj_host_alloc) and ordinary memory."""
    rng = random.Random(5)
    alphabet = b"abcx \nregxp"
    patterns = [b"regexp", b"[a-c]+x", b"x*", b"^", b"$", b"^a.*x$", b"(ab|bc)+", b"a[^x]*x"]
    n_sep_free = 0
    for rx in patterns:
        p = rj.Program(rx)
        sep = p.batch_separator()
        n_sep_free += sep < 0
        use = sep if sep >= 0 else 0
        for trial in range(4):
            k = rng.choice([1, 2, 7, 40])
            texts = []
            for _ in range(k):
                n = rng.choice([0, 0, 1, 5, 17, 300, 5000])
                t = bytes(rng.choice(alphabet) for _ in range(n))
                if sep >= 0:
                    t = t.replace(bytes([sep]), b"q")
                texts.append(t)
            lead = rng.choice([0, 0, 1, 5])
            buf, offsets, sizes = _layout(texts, use, lead, rng.choice([1, 3]), rng)
            want = [oracle.match_all(rx, t) for t in texts]
            got = _packed(rj, p, buf, offsets, sizes, pinned=bool(trial & 1))
            assert got == want, (rx, trial, lead, sizes[:8])
            assert p.match_all_batch(texts) == want, (rx, trial)
    assert n_sep_free == 0 or n_sep_free < len(patterns)


def test_packed_pattern_without_separator_goes_text_by_text(rj, oracle):
    """`[^a]*$` consumes every byte value but `a`... and `(.|\\n|\\r)*x` every one: rj_batch_separator is -1 for a pattern that
    leaves no byte free; the packed call still answers, text by text."""
    cands = [b"[^a]+|a+", b".*|[\n\r]+"]
    found = None
    for rx in cands:
        p = rj.Program(rx)
        if p.batch_separator() < 0:
            found = (rx, p)
            break
    assert found is not None, "no pattern without a separator among the candidates"
    rx, p = found
    rng = random.Random(8)
    texts = [bytes(rng.choice(b"ab\nq") for _ in range(n)) for n in (0, 3, 50, 1000, 1)]
    buf, offsets, sizes = _layout(texts, 0, 2, 2, rng)
    assert _packed(rj, p, buf, offsets, sizes, pinned=False) == [oracle.match_all(rx, t) for t in texts]


def test_packed_argument_errors(rj):
    """A last text without a byte behind it, overlapping texts and null arguments are RJ_BAD_ARGUMENT, not a crash."""
    p = rj.Program(b"abc")
    lib = rj.load_library()
    buf = ctypes.create_string_buffer(b"abc\0abc", 7)
    with pytest.raises(rj.RejitError) as e:
        p.match_all_packed(ctypes.addressof(buf), [0, 4], [3, 3], 7)       # the last text ends at total_bytes: no separator
    assert e.value.status == -2 or "separator" in e.value.message or "overlaps" in e.value.message
    with pytest.raises(rj.RejitError):
        p.match_all_packed(ctypes.addressof(buf), [0, 2], [3, 3], 8)       # text 0 runs into text 1
    counts = (ctypes.c_uint64 * 2)()
    rc = lib.rj_match_all_packed(p._h, None, None, None, 2, 8, counts, None)
    assert rc < 0
    assert lib.rj_match_all_packed(p._h, None, None, None, 0, 0, None, None) == 0   # an empty batch is no error


def test_scan_start_finish_two_in_flight(rj, oracle):
    """rj_scan_start / rj_scan_finish on two rj_scan objects used alternately over three texts: step by step the counts and
    spans of rj_scan_run -- windows, floating-window and dense patterns."""
    import torch
    from rejit_amd import workloads as W
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    sizes = (70000, 1 << 20, 300001)
    texts = []
    for i, n in enumerate(sizes):
        t = W.random_ascii_numpy(n, seed=31 + i)
        W.plant(t, W.plant_offsets(n, 6, 25, seed=i, boundaries=[1024, 65536]), b"regexp")
        texts.append(torch.from_numpy(t).to(dev))
    for rx in (b"regexp", b"[a-f]+[0-9]", b"re(ge|xx)xp"):
        p = rj.Program(rx)
        ref = rj.Scan(p)
        want = []
        for t in texts:
            c = ref.run(t.data_ptr(), t.numel(), stream=st)
            want.append((c, ref.spans()))
        if rx == b"regexp":
            assert want[0][1] == oracle.match_all(rx, texts[0].cpu().numpy().tobytes())
        a, b = rj.Scan(p), rj.Scan(p)
        objs = [a, b]
        steps = 9
        for k in range(steps):
            j = k & 1
            if k >= 2:
                c = objs[j].finish()
                assert (c, objs[j].spans()) == want[(k - 2) % 3], (rx, k)
            objs[j].start(texts[k % 3].data_ptr(), texts[k % 3].numel(), stream=st)
        for k in (steps - 2, steps - 1):
            c = objs[k & 1].finish()
            assert (c, objs[k & 1].spans()) == want[k % 3], (rx, k)


# ------------------------------------------------------------------------------------------------ samples/jrep_gpu
def _tree(base, rng, n_files=60, terminated=False):
    words = [b"int", b"regexp", b"return", b"for (;;)", b"x = y + 1;", b"// a comment", b"regexps", b"char* s", b"", b"}"]
    os.makedirs(base, exist_ok=True)
    for i in range(n_files):
        d = os.path.join(base, "d%d" % (i % 5), "sub%d" % (i % 3))
        os.makedirs(d, exist_ok=True)
        lines = []
        for _ in range(rng.choice([0, 1, 3, 40, 400])):
            lines.append(b" ".join(rng.choice(words) for _ in range(rng.randint(0, 6))))
        data = b"\n".join(lines)
        if lines and (terminated or rng.random() < 0.8):
            data += b"\n"
        with open(os.path.join(d, "f%03d.c" % i), "wb") as fh:
            fh.write(data)


def _run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, timeout=300)
    assert r.returncode in (0, 1), (cmd, r.returncode, r.stderr.decode()[-400:])
    return r.returncode, r.stdout


def test_native_jrep_equals_gnu_grep_and_reference_jrep(rj, tmp_path):
    """samples/jrep_gpu (C++ over the C ABI: batches of files per device pass) prints what GNU grep prints -- plain,
    -n, -H, context (-A/-B/-C), --count, any number of reader threads -- and what the real reference's jrep prints
    (oracle/_ref/jrep_ref: the reference's own sample on the reference's own library), exit status included."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "samples")], stdout=subprocess.DEVNULL)
    native = os.path.join(ROOT, "samples", "jrep_gpu")
    assert os.path.exists(native)
    base = str(tmp_path / "tree")
    _tree(base, random.Random(12))
    # the reference's jrep glues an unterminated last line to the next file's output (no line break of its own): it is
    # compared on a tree whose files all end in a line break, where it prints what GNU grep prints
    base_t = str(tmp_path / "tree_terminated")
    _tree(base_t, random.Random(13), terminated=True)
    grep = shutil.which("grep")
    ref = os.path.join(ROOT, "oracle", "_ref", "jrep_ref")
    checked = 0
    for pattern in ("regexp", "regexps|return", "nothing_matches_this"):
        for opts in (["-R", "-H", "-n"], ["-R", "-H"], ["-R", "-H", "-n", "-A2"], ["-R", "-H", "-n", "-B1"], ["-R", "-H", "-n", "-C3"],
                     ["-R", "-H", "-n", "-j3"], ["-R", "-H", "-n", "-j16"]):
            rc, out = _run([native] + opts + [pattern, "."], base)
            if grep and not any(o.startswith("-j") for o in opts):
                grc, gout = _run([grep, "-E"] + opts + [pattern, "."], base)
                # (files are visited in walk order, not grep's: compare as sorted lines, like tools/jrep_compare.py; grep
                # also puts a `--` between the context groups of DIFFERENT files, the reference's jrep format does not)
                strip = lambda o: sorted(l for l in o.splitlines() if l != b"--")
                assert strip(out) == strip(gout), (pattern, opts)
                assert rc == grc, (pattern, opts)
                checked += 1
            if os.path.exists(ref) and opts in (["-R", "-H", "-n"], ["-R", "-H", "-n", "-j3"], ["-R", "-H", "-n", "-j16"]):
                rrc, rout = _run([ref, "-R", "-H", "-n", pattern, "."], base_t)
                _, tout = _run([native] + opts + [pattern, "."], base_t)
                assert sorted(tout.splitlines()) == sorted(rout.splitlines()), (pattern, opts)
                checked += 1
        rc, out = _run([native, "-R", "--count", pattern, "."], base)
        if grep:
            # grep -c counts LINES with a match; the native sample's --count counts matches per file with a match:
            # compare the set of files
            _, gout = _run([grep, "-E", "-R", "-l", pattern, "."], base)
            assert sorted(l.rsplit(b":", 1)[0] for l in out.splitlines()) == sorted(gout.splitlines()), pattern
            checked += 1
    assert checked >= 6, "neither GNU grep nor oracle/_ref/jrep_ref on this box"


def test_kernel_timing_switch(rj, oracle):
    """rj_scan_set_timing / rj_multi_set_timing / rj_set_default_timing: without the scan kernel's start event the answers
    are the same and scan_ms reads 0 (the C default; the Python binding switches the default on when it loads the library)."""
    import torch
    from rejit_amd import workloads as W
    lib = rj.load_library()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    n = 3 << 20
    t = W.random_ascii_numpy(n, seed=77)
    W.plant(t, W.plant_offsets(n, 6, 40, seed=3, boundaries=[65536]), b"regexp")
    d = torch.from_numpy(t).to(dev)
    for rx in (b"regexp", b"[a-f]+[0-9]", b"^"):
        p = rj.Program(rx)
        a, b = rj.Scan(p), rj.Scan(p)
        b.set_timing(False)
        ca, cb = a.run(d.data_ptr(), n, stream=st), b.run(d.data_ptr(), n, stream=st)
        assert ca == cb and a.spans() == b.spans(), rx
        assert a.stats()["scan_ms"] > 0 and b.stats()["scan_ms"] == 0, rx
        if rx == b"regexp":
            assert a.spans() == oracle.match_all(rx, t.tobytes())
    prev = lib.rj_set_default_timing(0)
    try:
        assert prev == 1                      # (the binding's default)
        c = rj.Scan(rj.Program(b"regexp"))
        c.run(d.data_ptr(), n, stream=st)
        assert c.stats()["scan_ms"] == 0
    finally:
        lib.rj_set_default_timing(prev)
    progs = [rj.Program(x) for x in W.REGEXDNA_PATTERNS]
    f = W.fasta_stripped_torch(200_000, dev)
    m1, m2 = rj.MultiScan(progs), rj.MultiScan(progs)
    m2.set_timing(False)
    c1 = m1.run(f.data_ptr(), f.numel(), stream=st)
    c2 = m2.run(f.data_ptr(), f.numel(), stream=st)
    assert c1 == c2 and m1.scan_ms() > 0 and m2.scan_ms() == 0
    m2.set_tail_stream(True)
    m2.start(f.data_ptr(), f.numel(), stream=st)
    assert m2.finish() == c1
