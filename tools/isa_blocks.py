#!/usr/bin/env python3
"""CPU only: the instruction mix of a kernel per basic block, from the compiler's assembly (hipcc -S, gfx950).
For a kernel that is bound by VALU issue -- scan_dense_walk: SQ_INSTS_VALU x 4 cycles = its duration x 1024 SIMDs --
the v_* count of the blocks on the chunk loop's path IS the time; count before and after a change.
usage: isa_blocks.py <source.hip | file.s> <substring of the kernel's (demangled or mangled) name> [min VALU per block to list]
       e.g. isa_blocks.py rejit_amd/csrc/kernels.hip 'scan_dense_walk<1, false, 4>' 5"""
import os, re, subprocess, sys, tempfile

src, want = sys.argv[1], sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if src.endswith(".s"):
    text = open(src).read()
else:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
lines = text.split("\n")
labels = [(i, l[:-1].split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
names = subprocess.run(["c++filt"], input="\n".join(n for _, n in labels), capture_output=True, text=True).stdout.splitlines()
hits = [(i, m, d) for (i, m), d in zip(labels, names) if want in d or want in m]
if not hits:
    sys.exit("no kernel matches; candidates:\n  " + "\n  ".join(sorted(set(names))[:60]))
start, mangled, demangled = hits[0]
end = start
while "s_endpgm" not in lines[end]:
    end += 1
print("#", demangled)
blocks, cur = [], None
for l in lines[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or cur is None:
        cur = {"name": m.group(1) if m else "entry", "v": 0, "s": 0, "lds": 0, "mem": 0, "br": []}
        blocks.append(cur)
        if m:
            continue
    t = l.strip()
    if not t or t[0] in ";.":
        continue
    op = t.split()[0]
    if op.startswith("v_"): cur["v"] += 1
    elif op.startswith("s_"): cur["s"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.split("_")[0] in ("global", "buffer", "flat", "scratch"): cur["mem"] += 1
    if op.startswith(("s_cbranch", "s_branch")):
        cur["br"].append(op[2:] + " " + t.split()[1])
print("# total: VALU %d  SALU %d  LDS %d  memory %d  in %d blocks (a block may hold early exits: see its branches)" %
      (sum(b["v"] for b in blocks), sum(b["s"] for b in blocks), sum(b["lds"] for b in blocks), sum(b["mem"] for b in blocks), len(blocks)))
print(f"{'block':12s} {'valu':>5s} {'salu':>5s} {'lds':>4s} {'mem':>4s}  branches")
for b in blocks:
    if b["v"] >= thr:
        print(f"{b['name']:12s} {b['v']:5d} {b['s']:5d} {b['lds']:4d} {b['mem']:4d}  {', '.join(b['br'][:5])}")
