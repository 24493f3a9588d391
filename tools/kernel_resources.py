#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS figures of every HIP source, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950; no GPU needed).
usage: kernel_resources.py [source.hip ...] > profiles/<tag>_kernel_resources.txt"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from rejit_amd import api

srcs = sys.argv[1:] or [s for s in api.SOURCES if s.endswith(".hip")]
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage, per kernel: VGPRs, SGPRs, SGPR spills (to VGPR lanes),")
print("# VGPR spills and scratch bytes per lane (memory), occupancy in waves per SIMD, static LDS bytes per workgroup")
print(f"{'kernel':72s} {'vgpr':>5s} {'sgpr':>5s} {'s-spill':>7s} {'v-spill':>7s} {'scratch':>7s} {'occ':>3s} {'lds':>6s}")
for src in srcs:
    path = src if os.path.isabs(src) else os.path.join(api.CSRC, src)
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Rpass-analysis=kernel-resource-usage", "-c", path,
                            "-o", os.path.join(tmp, "o.o")], capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-2000:])
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    names = subprocess.run(["c++filt"], input="\n".join(b.split(" ")[0] for b in blocks), capture_output=True, text=True).stdout.splitlines()
    print(f"## {os.path.basename(path)}")
    for b, name in zip(blocks, names):
        def g(key):
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("rejit_amd::", "").replace("void ", "")
        scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{short[:72]:72s} {g('VGPRs'):5d} {g('TotalSGPRs'):5d} {g('SGPRs Spill'):7d} {g('VGPRs Spill'):7d} {scratch:7d} {occ:3d} {lds:6d}")
