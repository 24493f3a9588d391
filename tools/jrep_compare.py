#!/usr/bin/env python3
"""grep over a synthetic source tree, four ways, timed in one script (run on the GPU box):
  1. the reference's jrep on the reference's own library (oracle/_ref/jrep_ref), 1 thread and -j <cores>
  2. the reference's jrep UNCHANGED on librejit_hip.so (oracle/_ref/jrep_hip): one MatchAll per file
  3. samples/jrep_gpu.py: whole batches of files per device pass (rj_match_all_batch)
usage: jrep_compare.py [n_files] [dir]"""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # synthetic_tree

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
base = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="jrep_tree_")
files = bench.synthetic_tree(n_files, 4242)
total = 0
for i, data in enumerate(files):
    d = os.path.join(base, "d%03d" % (i // 200))
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "f%05d.c" % i), "wb") as fh:
        fh.write(data)
    total += len(data)
os.makedirs(os.path.join(base, "..", "jrep_empty_dir"), exist_ok=True)
print("tree: %d files, %.1f MB under %s" % (n_files, total / 1e6, base), flush=True)
ref = os.path.join(ROOT, "oracle", "_ref")
cores = os.cpu_count() or 1


def run(label, cmd, repeat=3):
    best, out = None, b""
    for _ in range(repeat):
        t0 = time.perf_counter()
        r = subprocess.run(cmd, cwd=base, capture_output=True)
        dt = time.perf_counter() - t0
        if r.returncode not in (0, 1):
            print("%-64s FAILED rc=%d %s" % (label, r.returncode, r.stderr.decode()[-300:]))
            return None
        out = r.stdout
        best = dt if best is None else min(best, dt)
    print("%-64s %8.1f ms  %7.2f GB/s  (%d output lines)" % (label, best * 1e3, total / best / 1e9, out.count(b"\n")), flush=True)
    return sorted(out.splitlines())


outs = {}
outs["ref1"] = run("reference jrep, reference library, 1 thread", [os.path.join(ref, "jrep_ref"), "-R", "-H", "-n", "regexp", "."])
# (argp: the optional argument of -j must be attached -- `-j 8` makes "8" the pattern)
outs["refj"] = run("reference jrep, reference library, -j8", [os.path.join(ref, "jrep_ref"), "-R", "-H", "-n", "-j8", "regexp", "."])
outs["hip1"] = run("reference jrep UNCHANGED on librejit_hip.so, 1 thread", [os.path.join(ref, "jrep_hip"), "-R", "-H", "-n", "regexp", "."])
for j in (8, 32, 128):
    outs["hipj%d" % j] = run("reference jrep UNCHANGED on librejit_hip.so, -j%d (calls combined in the library)" % j,
                             [os.path.join(ref, "jrep_hip"), "-R", "-H", "-n", "-j%d" % j, "regexp", "."])
run("  (start-up alone: the same binary over an empty directory)", [os.path.join(ref, "jrep_hip"), "-R", "-H", "-n", "regexp", os.path.join(base, "..", "jrep_empty_dir")])
native = os.path.join(ROOT, "samples", "jrep_gpu")
if os.path.exists(native):
    outs["native"] = run("samples/jrep_gpu (C++ over the C ABI: batches per device pass)", [native, "-R", "-H", "-n", "regexp", "."])
    run("  (start-up alone: the same binary over an empty directory)", [native, "-R", "-H", "-n", "regexp", os.path.join(base, "..", "jrep_empty_dir")])
outs["batch"] = run("samples/jrep_gpu.py (rj_match_all_batch, whole batches per pass)", [sys.executable, os.path.join(ROOT, "samples", "jrep_gpu.py"), "-R", "-H", "-n", "regexp", "."], repeat=2)
# the library alone on the same files, already in memory: the pattern's batch call + the line-table batch call
# over the files with matches (what a C++ caller of rj_match_all_batch pays; no file I/O, no formatting)
import rejit_amd
prog, sol = rejit_amd.Program(b"regexp"), rejit_amd.Program(b"^")
for _ in range(2):
    t0 = time.perf_counter()
    res = prog.match_all_batch(files)
    hit = [f for f, r in zip(files, res) if r]
    lines = sol.match_all_batch(hit) if hit else []
    dt = time.perf_counter() - t0
print("%-64s %8.1f ms  %7.2f GB/s  (%d files with matches; through the ctypes binding)" % ("rj_match_all_batch over the files in memory + line tables", dt * 1e3, total / dt / 1e9, len(hit)), flush=True)
if shutil.which("grep"):
    outs["grep"] = run("GNU grep -R -H -n", ["grep", "-R", "-H", "-n", "regexp", "."])
want = outs.get("ref1")
for k, v in outs.items():
    if v is not None and want is not None:
        print("  output of %-7s == reference jrep (sorted lines): %s" % (k, v == want))
if len(sys.argv) <= 2:
    shutil.rmtree(base, ignore_errors=True)
