#!/usr/bin/env python3
"""Turn the gpurun_out/prof_<tag>_* files written by tools/profile_round.sh into the committed
profiles/ files.  usage: collect_profiles.py <tag>"""
import json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda name: os.path.join(root, "gpurun_out", f"prof_{tag}_{name}")
hdr = """# rocprofv3 PMC passes, one counter per pass (MI355X, gfx950, ROCm 7.2), run by tools/profile_round.sh; command per pass:
#   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
#   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
# Unit: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM):
# FETCH_SIZE counts a wide coalesced 16-B/lane stream at exactly 1/2 -> HBM read bytes = FETCH_SIZE*1024*2.
# WRITE_SIZE is uncalibrated on gfx950 (guide); it is listed raw and is <0.2% of the read side here.

"""
f, w = open(g("pmc_fetch.txt")).read(), open(g("pmc_write.txt")).read()
open(os.path.join(root, "profiles", "r01_pmc_hbm_traffic.txt"), "w").write(
    hdr + "## FETCH_SIZE (KiB per launch)\n" + f + "\n## WRITE_SIZE (KiB per launch, raw)\n" + w)


def avg(txt, prefix):
    for line in txt.splitlines():
        if line.startswith(prefix):
            return float(line[60:].split()[1])
    raise KeyError(prefix)


j = {"source": "profiles/r01_pmc_hbm_traffic.txt",
     "regexdna": {"kernel": "scan_windows<2,true,true,false,true>", "fasta_n": 50000000,
                  "hbm_read_bytes_per_launch": avg(f, "scan_windows<2, true, true, false, true>") * 1024 * 2},
     "literal": {"kernel": "scan_windows<1,true,false,true,false>", "bytes": 5000000000,
                 "hbm_read_bytes_per_launch": avg(f, "scan_windows<1, true, false, true, false>") * 1024 * 2}}
json.dump(j, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
ks = open(g("kernel_stats.txt")).read()
prof_line = open(g("bench_profiled.json")).read().strip().splitlines()[-1]
open(os.path.join(root, "profiles", "r01_bench_kernel_stats.txt"), "w").write(
    "# rocprofv3 --kernel-trace --stats -- python bench.py --no-extra --no-cpu-baseline   (MI355X)\n"
    "# = the headline run alone (default K/W, the extras and the CPU sample left out so that the per-kernel\n"
    "# averages are those of the timed region); summarised from the rocpd database by tools/prof_summary.py.\n"
    "# The un-profiled bench line (profiles/r01_bench_line.json) reports roofline.avg_launch_ms, which agrees\n"
    "# with scan_windows avg_us below; under the profiler the event-based time reads ~7% higher:\n# " + prof_line + "\n" + ks)
open(os.path.join(root, "profiles", "r01_bench_line.json"), "w").write(open(g("bench.json")).read().strip().splitlines()[-1] + "\n")
d = json.load(open(os.path.join(root, "profiles", "r01_bench_line.json")))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"])
for k in ("overlapped", "fused"):
    print(k, d.get(k))
for k in ("literal_scan", "complex_scan"):
    print(k, d[k]["value"], d[k]["latency_ms"], d[k]["roofline"]["frac"], d[k]["roofline"]["traffic"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
