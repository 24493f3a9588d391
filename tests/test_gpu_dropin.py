"""Drop-in proof on the GPU: the reference's OWN callers -- sample/regexdna.cc and
sample/jrep.cc, compiled UNCHANGED against our include/rejit.h and linked against our
librejit_hip.so (oracle/Makefile target `callers`, built in the build container) -- produce
the canonical regex-dna output and grep-identical results."""
import os
import shutil
import subprocess

import pytest

import vectors as V
from rejit_amd import workloads as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REGEXDNA = os.path.join(ROOT, "oracle", "_ref", "regexdna_hip")
JREP = os.path.join(ROOT, "oracle", "_ref", "jrep_hip")


@pytest.mark.skipif(not os.path.exists(REGEXDNA), reason="oracle/_ref/regexdna_hip not built")
@pytest.mark.parametrize("nf", [1000, 50000])
def test_reference_regexdna_sample_runs_on_our_library(tmp_path, nf):
    g = V.bench()["regexdna"][str(nf)]
    fasta = tmp_path / "in.fasta"
    fasta.write_bytes(W.fasta_raw_numpy(nf).tobytes())
    with open(fasta, "rb") as f:
        out = subprocess.run([REGEXDNA], stdin=f, capture_output=True, timeout=300, check=True).stdout.decode()
    lines = out.strip().split("\n")
    counts = [int(l.rsplit(" ", 1)[1]) for l in lines[:9]]
    assert [l.rsplit(" ", 1)[0] for l in lines[:9]] == W.REGEXDNA_PATTERNS
    assert counts == [p["count"] for p in g["patterns"]]
    sizes = [int(x) for x in lines[-3:]]
    assert sizes == [g["raw_size"], g["stripped_size"], g["replaced_size"]]
    if nf == 50000:   # the canonical Benchmarks-Game output
        assert counts == [3, 12, 43, 27, 58, 16, 15, 18, 20] and sizes == [508411, 500000, 668262]


@pytest.mark.skipif(not os.path.exists(JREP) or shutil.which("grep") is None, reason="jrep_hip or grep missing")
def test_reference_jrep_sample_equals_grep(tmp_path):
    import random
    rng = random.Random(3)
    words = ["regexp", "alpha", "beta", "gamma", "int", "return", "for", "while", "x", "y", "regex", "exp"]
    for d in ("a", "a/b", "c"):
        os.makedirs(tmp_path / d, exist_ok=True)
    for i in range(40):
        sub = rng.choice(["a", "a/b", "c", "."])
        lines = [" ".join(rng.choice(words) for _ in range(rng.randint(1, 9))) for _ in range(rng.randint(1, 60))]
        (tmp_path / sub / f"f{i}.txt").write_text("\n".join(lines) + "\n")
    for pattern in ("regexp", "gamma beta", "whil"):
        ours = subprocess.run([JREP, "-R", "-H", "-n", pattern, "."], cwd=tmp_path, capture_output=True, timeout=300).stdout
        ref = subprocess.run(["grep", "-R", "-H", "-n", pattern, "."], cwd=tmp_path, capture_output=True).stdout
        assert sorted(ours.splitlines()) == sorted(ref.splitlines()), pattern
        assert len(ref.splitlines()) > 0


@pytest.mark.skipif(shutil.which("grep") is None, reason="grep missing")
def test_own_jrep_counterpart_equals_grep_and_reference_jrep(tmp_path):
    """samples/jrep_gpu.py (whole batches of files in one device pass, rj_match_all_batch) prints
    what GNU grep prints for literal patterns, and what the reference's own jrep prints on our
    library for an alternation."""
    import random
    import sys
    rng = random.Random(5)
    words = ["regexp", "alpha", "beta", "gamma", "int", "return", "for", "while", "x", "y", "regex", "exp"]
    for d in ("a", "a/b", "c"):
        os.makedirs(tmp_path / d, exist_ok=True)
    for i in range(120):
        sub = rng.choice(["a", "a/b", "c", "."])
        lines = [" ".join(rng.choice(words) for _ in range(rng.randint(0, 9))) for _ in range(rng.randint(0, 60))]
        (tmp_path / sub / f"f{i}.txt").write_text("\n".join(lines) + ("\n" if i % 7 or sub == "c" else ""))
    sample = os.path.join(ROOT, "samples", "jrep_gpu.py")
    for pattern, extra in (("regexp", []), ("gamma beta", []), ("whil", ["-C", "1"]), ("x y", ["-A", "2"])):
        ours = subprocess.run([sys.executable, sample, "-R", "-H", "-n", *extra, "--batch-mib", "1", pattern, "."], cwd=tmp_path,
                              capture_output=True, timeout=600, check=True).stdout
        ref = subprocess.run(["grep", "-R", "-H", "-n", *extra, pattern, "."], cwd=tmp_path, capture_output=True).stdout
        if extra:   # with context the file order matters (group separators): compare per file
            def per_file(blob):
                d = {}
                for line in blob.splitlines():
                    if line == b"--":
                        continue
                    d.setdefault(line.split(b"-")[0].split(b":")[0], []).append(line)
                return d
            assert per_file(ours) == per_file(ref), pattern
        else:
            assert sorted(ours.splitlines()) == sorted(ref.splitlines()), pattern
        assert len(ref.splitlines()) > 0
    if os.path.exists(JREP):
        # (the reference's jrep prints a last line without a line break as it is, so its output glues
        # it to the next file's first line; grep and jrep_gpu.py end it -- compare where files end in \n)
        pattern = "(regexp|gamma) (x|y)"
        ours = subprocess.run([sys.executable, sample, "-R", "-H", "-n", pattern, "c"], cwd=tmp_path, capture_output=True,
                              timeout=600, check=True).stdout
        theirs = subprocess.run([JREP, "-R", "-H", "-n", pattern, "c"], cwd=tmp_path, capture_output=True, timeout=300).stdout
        assert sorted(ours.splitlines()) == sorted(theirs.splitlines()) and len(ours.splitlines()) > 0


@pytest.mark.skipif(shutil.which("grep") is None, reason="grep missing")
def test_own_jrep_counterpart_two_ranks(tmp_path):
    """The file-sharded path of samples/jrep_gpu.py (BASELINE C5 shape): two ranks (gloo, both on
    cuda:0 -- the GPU box has one GPU) split the files, rank 0 prints the gathered output; as a set of
    lines it equals GNU grep."""
    import random
    import socket
    import sys
    rng = random.Random(8)
    words = ["regexp", "alpha", "beta", "gamma", "int", "return", "x", "y"]
    os.makedirs(tmp_path / "d", exist_ok=True)
    for i in range(60):
        lines = [" ".join(rng.choice(words) for _ in range(rng.randint(0, 8))) for _ in range(rng.randint(0, 80))]
        (tmp_path / ("d" if i % 2 else ".") / f"g{i}.txt").write_text("\n".join(lines) + "\n")
    sample = os.path.join(ROOT, "samples", "jrep_gpu.py")
    for attempt in range(2):   # (a rendezvous on a just-freed port can fail once in a while)
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ours = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                               "--master-addr", "127.0.0.1", "--master-port", str(port), sample, "--backend", "gloo",
                               "--same-device", "-o", str(tmp_path / "out.txt"), "-R", "-H", "-n", "regexp", "."],
                              cwd=tmp_path, capture_output=True, timeout=900)
        if ours.returncode == 0:
            break
    assert ours.returncode == 0, ours.stderr.decode()[-2000:]
    ref = subprocess.run(["grep", "-R", "-H", "-n", "--exclude=out.txt", "regexp", "."], cwd=tmp_path, capture_output=True).stdout
    # (the launcher shares one stdout between the ranks and gloo's rendezvous log: the result goes to a file)
    got = (tmp_path / "out.txt").read_bytes().splitlines()
    assert sorted(got) == sorted(ref.splitlines()) and len(ref.splitlines()) > 0
    # ... and in the order of a single rank (every file's output travels with its place in the walk)
    one = subprocess.run([sys.executable, sample, "-o", str(tmp_path / "one.txt"), "-R", "-H", "-n", "regexp", "."], cwd=tmp_path,
                         capture_output=True, timeout=600)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    single = [l for l in (tmp_path / "one.txt").read_bytes().splitlines() if not l.startswith((b"./out.txt", b"./one.txt"))]
    assert [l for l in got if not l.startswith((b"./out.txt", b"./one.txt"))] == single


def test_own_regexdna_counterpart(tmp_path):
    """samples/regexdna_gpu.py (text resident in HBM for the whole program) prints the same
    output as the reference's sample on the same input."""
    import sys
    g = V.bench()["regexdna"]["50000"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "samples", "regexdna_gpu.py"), "--n", "50000"],
                         capture_output=True, timeout=600, check=True).stdout.decode()
    lines = out.strip().split("\n")
    assert [int(l.rsplit(" ", 1)[1]) for l in lines[:9]] == [3, 12, 43, 27, 58, 16, 15, 18, 20]
    assert [int(x) for x in lines[-3:]] == [508411, 500000, 668262]


@pytest.mark.parametrize("nf,mode", [(50000, "--timing"), (50000, "--serial"), (5000000, "--timing")])
def test_native_regexdna_counterpart(tmp_path, nf, mode):
    """samples/regexdna_gpu (C++ over the C ABI: one upload, strip / nine counts in one pass / eleven replacements all on the
    device, the counts and the replacements in flight together) prints byte for byte what the reference's own program
    prints on this library (oracle/_ref/regexdna_hip: sample/regexdna.cc unchanged) and the canonical answers."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "samples")], stdout=subprocess.DEVNULL)
    fasta = tmp_path / "in.fasta"
    fasta.write_bytes(W.fasta_raw_numpy(nf).tobytes())
    with open(fasta, "rb") as f:
        r = subprocess.run([os.path.join(ROOT, "samples", "regexdna_gpu"), mode], stdin=f, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode()
    lines = out.strip().split("\n")
    assert [l.rsplit(" ", 1)[0] for l in lines[:9]] == W.REGEXDNA_PATTERNS
    if nf == 50000:
        assert [int(l.rsplit(" ", 1)[1]) for l in lines[:9]] == [3, 12, 43, 27, 58, 16, 15, 18, 20]
        assert [int(x) for x in lines[-3:]] == [508411, 500000, 668262]
    exe = os.path.join(ROOT, "oracle", "_ref", "regexdna_hip")
    if os.path.exists(exe):
        with open(fasta, "rb") as f:
            want = subprocess.run([exe], stdin=f, capture_output=True, timeout=900, check=True).stdout.decode()
        assert out == want


REF_DIR = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "regexdna_mt_hip")), reason="oracle/_ref/regexdna_mt_hip not built")
def test_reference_regexdna_multithread_sample(tmp_path):
    """sample/regexdna-multithread.cc unchanged: a Regej per thread per call, hardware_concurrency()
    threads hammering one library (reference sample/regexdna-multithread.cc:65-78, 117-167)."""
    g = V.bench()["regexdna"]["50000"]
    fasta = tmp_path / "in.fasta"
    fasta.write_bytes(W.fasta_raw_numpy(50000).tobytes())
    with open(fasta, "rb") as f:
        out = subprocess.run([os.path.join(REF_DIR, "regexdna_mt_hip")], stdin=f, capture_output=True, timeout=600, check=True).stdout.decode()
    lines = out.strip().split("\n")
    assert [l.rsplit(" ", 1)[0] for l in lines[:9]] == W.REGEXDNA_PATTERNS
    assert [int(l.rsplit(" ", 1)[1]) for l in lines[:9]] == [3, 12, 43, 27, 58, 16, 15, 18, 20]
    assert [int(x) for x in lines[-3:]] == [g["raw_size"], g["stripped_size"], g["replaced_size"]] == [508411, 500000, 668262]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "basic_hip")), reason="oracle/_ref/basic_hip not built")
def test_reference_basic_sample(tmp_path):
    """sample/basic.cc unchanged (MatchAll over a file named LICENCE through the string overload)."""
    from checkers import Oracle
    text = ("Copyright holders reserve every right.  Redistribution, conversion and extension of this\n"
            "section are subject to the conditions of the licence; no permission is implied.\n") * 7
    (tmp_path / "LICENCE").write_text(text)
    out = subprocess.run([os.path.join(REF_DIR, "basic_hip")], cwd=tmp_path, capture_output=True, timeout=120, check=True).stdout
    want = Oracle().match_all(b"(right|[ts]ion)", text.encode())
    assert b"Found %d matches." % len(want) in out, out
    # (basic.cc:44 passes a `const char*` where a string is expected: its Match pointers refer to a
    # temporary that is gone when it prints them -- the printed bytes are undefined with the reference
    # too, so only the count is checked)
    assert b"Printing the first 10:" in out


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "test_rejit_hip")), reason="oracle/_ref/test_rejit_hip not built")
def test_reference_test_program_passes():
    """tools/tests/test.cc unchanged -- the reference's ONLY test suite (282 macros, ~4.7 k checks of
    all four match types, 33 alignments each for the fast-forward cases) -- prints `success` on this
    library.  (The reference itself fails one of them with its default flags)"""
    r = subprocess.run([os.path.join(REF_DIR, "test_rejit_hip")], capture_output=True, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0 and out.strip().endswith("success"), out[-2000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "bench_rejit_hip")), reason="oracle/_ref/bench_rejit_hip not built")
def test_reference_bench_harness_runs():
    """tools/benchmarks/engines/{bench_engine.cc, rejit/engine.cc} unchanged: the harness the
    reference's published curves come from (bytes/s worst / amortised / best per text size)."""
    r = subprocess.run([os.path.join(REF_DIR, "bench_rejit_hip"), "regexp", "--size=1024,65536,4194304", "--iterations=5"],
                       capture_output=True, timeout=300, check=True)
    rows = [l.split() for l in r.stdout.decode().strip().split("\n")]
    assert rows[0][0] == "text_size" and [int(x[0]) for x in rows[1:]] == [1024, 65536, 4194304]
    assert all(float(v) > 0 for x in rows[1:] for v in x[1:])
