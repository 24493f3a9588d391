#!/usr/bin/env python3
"""plane_count.hip reads its code planes through the VGPR index mode (s_set_gpr_idx_*), which writes M0 behind the compiler's
back (M0 is reserved: an asm clobber is ignored).  This check compiles the file to ISA and fails when any instruction of the
plane_count kernels other than s_set_gpr_idx_* mentions m0, or when the indexed reads do not use the registers the planes were
pinned to (v48.. / v52.. / v56..).   usage: check_m0.py   (no GPU needed)"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "rejit_amd", "csrc", "plane_count.hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "pc.s")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
bad, kernels = [], 0
for m in re.finditer(r"^(_ZN9rejit_amd11plane_count[^\n:]*):.*?s_endpgm", text, re.S | re.M):
    kernels += 1
    inside = False
    for line in m.group(0).splitlines():
        ins = line.split(";")[0].strip()
        if ins.startswith("s_set_gpr_idx_on"):
            inside = True
        elif ins.startswith("s_set_gpr_idx_off"):
            inside = False
        elif re.search(r"\bm0\b", ins) and not ins.startswith("s_set_gpr_idx"):
            bad.append((m.group(1)[:60], ins))
        elif inside and ins.startswith("v_") and not re.search(r"\bv(48|52|56)\b", ins):
            bad.append((m.group(1)[:60], "indexed read without a pinned plane register: " + ins))
print("plane_count kernels checked: %d; offending instructions: %d" % (kernels, len(bad)))
for k, ins in bad[:20]:
    print(" ", k, ins)
sys.exit(1 if bad or kernels == 0 else 0)
