#!/usr/bin/env python3
"""Turn the gpurun_out/prof_<tag>_* files written by tools/profile_round.sh (and the probes run with it) into
the committed profiles/<tag>_* files and profiles/pmc_traffic.json.   usage: collect_profiles.py <tag>"""
import json, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda name: os.path.join(root, "gpurun_out", f"prof_{tag}_{name}")
P = lambda name: os.path.join(root, "profiles", f"{tag}_{name}")
hdr = f"""# rocprofv3 PMC passes, one counter group per pass (MI355X, gfx950, ROCm 7.2), run by tools/profile_round.sh; command per pass:
#   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --jrep-files 2000 --jrep-bytes 200000000
#   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- (the same)
# Unit: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM):
# FETCH_SIZE counts a wide coalesced 16-B/lane stream at exactly 1/2 -> HBM read bytes = FETCH_SIZE*1024*2.
# FETCH_SIZE also counts requests the 256 MiB Infinity Cache serves (same guide): for scan_windows_train
# (nine passes of every wave over its own 32 KB span) it shows the bytes REQUESTED from beyond L2, 0.98 x the
# 9 x 500 MB algorithmic bytes, not what HBM delivered.  WRITE_SIZE is uncalibrated on gfx950; listed raw.
# plane_scan<2> / scan_windows<..> run at two text sizes in this command (500 MB and 2.5 GB; 5 GB and 50 GB): min = the small one.

"""
f, w = open(g("pmc_fetch.txt")).read(), open(g("pmc_write.txt")).read()
lin = ""
if os.path.exists(g("pmc_fetch_linear.txt")):
    lin = ("\n## FETCH_SIZE (KiB per launch) of tools/linear_probe.py '[acgt]+' 'a.*b' (64 MiB = 65536 KiB single-run texts; x 2 for HBM bytes):\n"
           "## run_summary + run_emit read the text once each (round 5's carry scan: cs_emit_kernel 4 581 077 KiB, cs_local_chain_kernel 3 717 552 KiB per launch)\n"
           + open(g("pmc_fetch_linear.txt")).read())
open(P("pmc_hbm_traffic.txt"), "w").write(hdr + "## FETCH_SIZE (KiB per launch)\n" + f + "\n## WRITE_SIZE (KiB per launch, raw)\n" + w + lin)


def avg(txt, prefix):
    for line in txt.splitlines():
        if line.startswith(prefix):
            return float(line[60:].split()[1])
    raise KeyError(prefix)


def maybe(key, prefix, **extra):
    """a pmc_traffic.json entry when the kernel shows up in the FETCH_SIZE pass"""
    try:
        j[key] = dict(kernel=prefix, hbm_read_bytes_per_launch=avg(f, prefix) * 1024 * 2, **extra)
    except KeyError:
        pass


def avg_of(txt, prefix, expect_bytes):
    """kernels that run at several text sizes in one pass (plane_scan at 500 MB and 2.5 GB, small calls of the latency sweep):
    the group of launches (pmc_summary.py's last column) whose FETCH_SIZE is nearest to the text size"""
    for line in txt.splitlines():
        if line.startswith(prefix):
            groups = [float(x.split("x")[0]) for x in line[60:].split()[4:]]
            return min(groups, key=lambda kib: abs(kib * 2048 / expect_bytes - 1))
    raise KeyError(prefix)


j = {"source": f"profiles/{tag}_pmc_hbm_traffic.txt",
     "note": "HBM read bytes per launch = FETCH_SIZE (KiB) x 1024 x 2 (gfx950 counts a wide coalesced stream at 1/2, MI355X guide); kernels that "
             "run at two text sizes in the profiled command take the min / max launch"}
FASTA = 500000000  # bytes of the stripped 50 M-base FASTA input (workloads.fasta_stripped_size)
for key, prefix, nbytes, extra in (
        ("plane_count", "plane_count<ExactShape<2> >", FASTA, {"fasta_n": 50000000}),
        ("plane_count_2p5gb", "plane_count<ExactShape<2> >", 2500000000, {}),
        ("counts_general", "plane_count<GeneralShape<1, 16, false> >", 5000000000, {}),
        ("plane", "plane_count<ListShape<2> >", FASTA, {"fasta_n": 50000000}),
        ("plane_2p5gb", "plane_count<ListShape<2> >", 2500000000, {}),
        ("regexdna_single", "scan_windows<2, true, true, false, true>", FASTA, {"fasta_n": 50000000}),
        ("regexdna_single_2p5gb", "scan_windows<2, true, true, false, true>", 2500000000, {}),
        # (`regexp` is a 6-byte window: the MASKED instantiation; `abcdefgh` of the complex / behind patterns fills its 8 bytes)
        ("literal", "scan_windows<1, true, true, true, false>", 5000000000, {}),
        ("literal_50gb", "scan_windows<1, true, true, true, false>", 50000000000, {}),
        ("complex", "scan_windows<1, true, false, true, false>", 5000000000, {}),
        ("dense", "dense_streams<2, 2, false, false, true>", 5000000000, {}),
        ("dense_select", "dense_streams<3, 1, false, true, false>", 5000000000, {}),
        ("general", "plane_count<GeneralListShape<false> >", 5000000000, {}),
        ("line_table", "emit_assertions", 5000000000, {})):
    try:
        j[key] = dict(kernel=prefix, hbm_read_bytes_per_launch=avg_of(f, prefix, nbytes) * 1024 * 2, bytes=nbytes, **extra)
    except KeyError:
        pass
# VALU lane-operations per text byte of the issue-bound kernels: SQ_INSTS_VALU (wave instructions per launch, the SQ pass of
# tools/profile_round.sh: bench.py --no-big, so every kernel runs at ONE text size there) x 64 lanes / text bytes -- what
# bench.py's roofline_valu quotes (round 5 typed these into rejit_amd/__init__.py by hand)
try:
    sq = open(g("pmc_sq.txt")).read().split("## SQ_INSTS_VALU")[1].split("## ")[0]
    for key, prefix, nbytes in (("plane_count", "plane_count<ExactShape<2> >", FASTA), ("plane", "plane_count<ListShape<2> >", FASTA),
                                ("dense", "dense_streams<2, 2, false, false, true>", 5000000000), ("dense_select", "dense_streams<3, 1, false, true, false>", 5000000000),
                                ("counts_general", "plane_count<GeneralShape<1, 16, false> >", 5000000000), ("general", "plane_count<GeneralListShape<false> >", 5000000000)):
        ops = None
        for line in sq.splitlines():
            if line.startswith(prefix):
                # (the launch group nearest to what the kernel's instruction count per byte of the OTHER launches says is not known
                # here: take the largest group -- the full-size text; the small calls of the latency sweeps are far below it)
                groups = [float(x.split("x")[0]) for x in line[60:].split()[4:]]
                ops = max(groups) * 64 / nbytes
        if ops is not None and key in j:
            j[key]["valu_ops_per_text_byte"] = round(ops, 3)
            j[key]["valu_source"] = f"profiles/{tag}_pmc_sq_counters.txt: SQ_INSTS_VALU x 64 / {nbytes}"
except (OSError, IndexError):
    pass
# (behind: the same scan kernel as `complex`, over the same text)
if "complex" in j:
    j["behind"] = dict(j["complex"], note="same kernel and text as `complex`")
json.dump(j, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
prof_line = open(g("bench_profiled.json")).read().strip().splitlines()[-1]
open(P("bench_kernel_stats.txt"), "w").write(
    "# rocprofv3 --kernel-trace --stats -- python bench.py --no-extra --no-cpu-baseline   (MI355X)\n"
    "# = the headline run alone (default K/W; the extras and the CPU sample left out so that the per-kernel averages are\n"
    "# those of the timed region); summarised from the rocpd database by tools/rocpd_stats.py.  The un-profiled bench line\n"
    f"# (profiles/{tag}_bench_line.json) reports roofline.avg_launch_ms, which agrees with plane_count<ExactShape<2> > avg_us below\n"
    "# (under the profiler the dispatch-timestamp time reads a few % higher):\n# " + prof_line + "\n" + open(g("kernel_stats.txt")).read())
open(P("bench_full_kernel_stats.txt"), "w").write(
    "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 5 --jrep-files 5000 --jrep-bytes 500000000   (MI355X):\n"
    "# every kernel of the bench line's extras -- one-pass / train / per-pattern scans at 500 MB and 2.5 GB, literal at 5 and 50 GB, complex\n"
    "# (floating), behind, dense, line table (emit_assertions), batches, tails\n"
    + open(g("kernel_stats_full.txt")).read())
open(P("linear_path_kernel_stats.txt"), "w").write(
    "# rocprofv3 --kernel-trace --stats -- python tools/linear_probe.py   (MI355X): `[acgt]+` / `a.*b` 64 MiB single-line (round 6: the run\n"
    "# kernels run_summary / run_resolve / run_emit; until round 5 the carry scan cs_*), `x*` 16 MiB, `[ab]{40}c*` 4 MiB (carry scan / chain selection)\n"
    + "\n".join(l for l in open(g("linear_probe.txt")).read().splitlines() if " run " in l) + "\n" + open(g("kernel_stats_linear.txt")).read())
open(P("pmc_sq_counters.txt"), "w").write(
    "# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY\n"
    "#           --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-big   (MI355X)\n"
    "# per kernel: calls, avg / min / max of the counter per launch.  SQ_INSTS_VALU are WAVE instructions: / (text bytes / 1024) = per\n"
    "# 1-KiB chunk and wave = per 16 bytes and lane.\n" + open(g("pmc_sq.txt")).read())
for src, dst in (("dense_probe.txt", "dense_probe.txt"), ("jrep_compare.txt", "jrep_compare.txt"), ("bench_sizes.txt", "bench_sizes.txt"),
                 ("count_general_probe.txt", "count_general_probe.txt"), ("e2e_probe.txt", "e2e_probe.txt"), ("host_copy_probe.txt", "host_copy_probe.txt"), ("run_probe.txt", "run_probe.txt")):
    if os.path.exists(g(src)):
        shutil.copy(g(src), P(dst))
# the pair kernels (tools/probes/pair_profile.sh)
if os.path.exists(g("pair_probe.txt")):
    shutil.copy(g("pair_probe.txt"), P("pair_probe.txt"))
    open(P("pair_kernel_stats.txt"), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python tools/probes/pair_time.py 1024   (MI355X): `\"[^\"]*\"` (pair_*<1>) and `\"[^\"\\n]*\"` (pair_*<4>)\n"
        "# over 1 GiB of JSON-like text and 1 GiB of long strings; the at::native kernels build the texts\n" + open(g("pair_kernel_stats.txt")).read())
    open(P("pair_pmc_fetch.txt"), "w").write(
        "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace -- python tools/probes/pair_time.py 1024 (own passes): KiB per launch; x 2 for HBM\n"
        "# read bytes (gfx950: MI355X_MICROARCH.md) -- pair_summary reads the 1 GiB text once; pair_emit once more where tiles close a match\n"
        "## FETCH_SIZE\n" + open(g("pair_pmc_fetch.txt")).read() + "## WRITE_SIZE (raw)\n" + open(g("pair_pmc_write.txt")).read())
open(P("bench_line.json"), "w").write(open(g("bench.json")).read().strip().splitlines()[-1] + "\n")
d = json.load(open(P("bench_line.json")))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"])
