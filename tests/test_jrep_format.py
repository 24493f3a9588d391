"""samples/jrep_gpu.py: the host step after the path (offsets -> `file:line:text`, jrep.cc:300-400)
checked against GNU grep without a GPU.  The match offsets of the literal patterns come from
Python's `re` here -- scaffolding for the formatter only; the GPU end-to-end run is in
tests/test_gpu_dropin.py."""
import importlib.util
import io
import os
import random
import re
import shutil
import subprocess
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_sample():
    spec = importlib.util.spec_from_file_location("jrep_gpu", os.path.join(ROOT, "samples", "jrep_gpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(shutil.which("grep") is None, reason="GNU grep missing")
def test_line_output_equals_grep(tmp_path):
    m = load_sample()
    rng = random.Random(1)
    words = ["regexp", "alpha", "beta", "x", "regex", "exp", "gamma"]
    checked = 0
    for _ in range(150):
        lines = [" ".join(rng.choice(words) for _ in range(rng.randint(0, 5))) for _ in range(rng.randint(0, 30))]
        data = ("\n".join(lines) + ("\n" if rng.random() < 0.8 else "")).encode()
        pat = rng.choice(["regexp", "alpha beta", "x", "exp"])
        after, before = rng.choice([(0, 0), (1, 0), (0, 2), (2, 2), (3, 1)])
        matches = [(mm.start(), mm.end()) for mm in re.finditer(re.escape(pat).encode(), data)]
        if not matches:
            continue
        line_starts = [0] + [i + 1 for i, c in enumerate(data) if c == 10]   # what MatchAll("^") returns
        args = types.SimpleNamespace(with_filename=True, line_number=True, color=False, before=before, after=after)
        out = io.BytesIO()
        m.print_file(out, "f.txt", data, matches, line_starts, args)
        (tmp_path / "f.txt").write_bytes(data)
        cmd = ["grep", "-H", "-n"] + (["-A", str(after)] if after else []) + (["-B", str(before)] if before else []) + [pat, "f.txt"]
        ref = subprocess.run(cmd, cwd=tmp_path, capture_output=True).stdout
        assert out.getvalue() == ref, (pat, after, before, data)
        checked += 1
    assert checked > 80


def test_batches_split_by_size(tmp_path):
    m = load_sample()
    names = []
    for i in range(10):
        p = tmp_path / f"f{i}"
        p.write_bytes(b"x" * 1000)
        names.append(str(p))
    got = list(m.batches(names, 2500))
    assert [len(b) for b in got] == [3, 3, 3, 1]
    assert [n for b in got for n, _ in b] == names
