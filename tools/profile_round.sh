#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the bench line, kernel-trace stats of the headline run and of the full bench (incl.
# the 50 GB and 2.5 GB runs), and the PMC passes (own runs, --kernel-trace only -- never combined with other trace
# domains).  Summaries land in gpurun_out/prof_<tag>_*; tools/collect_profiles.py turns them into profiles/<tag>_*.
tag=${1:-r04}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out; mkdir -p $out
make -C $root/samples > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_full /tmp/prof_f /tmp/prof_w /tmp/prof_sq /tmp/prof_lin
python $root/bench.py > $out/prof_${tag}_bench.json 2> /tmp/b.log
# headline run only (no extras, no CPU sample): the per-kernel averages then are those of the timed region
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python $root/bench.py --no-extra --no-cpu-baseline > $out/prof_${tag}_bench_profiled.json 2> /tmp/kt.log
python $root/tools/rocpd_stats.py $(find /tmp/prof_kt -name "*.db" | head -1) 12 > $out/prof_${tag}_kernel_stats.txt
# every extra of the bench line, the 50 GB literal scan and the 2.5 GB runs included: one table
rocprofv3 --kernel-trace --stats -d /tmp/prof_full -o r -- python $root/bench.py --no-cpu-baseline --steps 5 --jrep-files 5000 --jrep-bytes 500000000 > /dev/null 2> /tmp/full.log
python $root/tools/rocpd_stats.py $(find /tmp/prof_full -name "*.db" | head -1) 56 split > $out/prof_${tag}_kernel_stats_full.txt
# the linear-time carry scan and the dense / class patterns, first and warm calls
rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o r -- python $root/tools/linear_probe.py > $out/prof_${tag}_linear_probe.txt 2> /tmp/lin.log
python $root/tools/rocpd_stats.py $(find /tmp/prof_lin -name "*.db" | head -1) 24 > $out/prof_${tag}_kernel_stats_linear.txt
# ... and what the run kernels fetch on `[acgt]+` / `a.*b` over 64 MiB (round 5's carry scan: 70-140 x the text)
rm -rf /tmp/prof_linf
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_linf -o r -- python $root/tools/linear_probe.py "[acgt]+" "a.*b" > /dev/null 2> /tmp/linf.log
python $root/tools/pmc_summary.py /tmp/prof_linf FETCH_SIZE > $out/prof_${tag}_pmc_fetch_linear.txt
# PMC passes: one counter group per run (the big runs included: no more `traffic: null`)
PMC_CMD="python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --jrep-files 2000 --jrep-bytes 200000000"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o r -- $PMC_CMD > /dev/null 2> /tmp/f.log
python $root/tools/pmc_summary.py /tmp/prof_f FETCH_SIZE > $out/prof_${tag}_pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o r -- $PMC_CMD > /dev/null 2> /tmp/w.log
python $root/tools/pmc_summary.py /tmp/prof_w WRITE_SIZE > $out/prof_${tag}_pmc_write.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_sq -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-big > /dev/null 2> /tmp/sq.log
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY; do
  echo "## $c"; python $root/tools/pmc_summary.py /tmp/prof_sq $c | head -16
done > $out/prof_${tag}_pmc_sq.txt
cd $root
python tools/jrep_compare.py 2>&1 | grep -v amdgpu.ids > $out/prof_${tag}_jrep_compare.txt
python tools/bench_sizes.py all > $out/prof_${tag}_bench_sizes.txt 2>/dev/null
python tools/dense_probe.py 1e9 > $out/prof_${tag}_dense_probe.txt 2>/dev/null
python tools/count_general_probe.py 1000000000 30 2>/dev/null | grep -v amdgpu.ids > $out/prof_${tag}_count_general_probe.txt
python tools/e2e_probe.py 50000000 3 2>/dev/null | grep -v amdgpu.ids > $out/prof_${tag}_e2e_probe.txt
python tools/probes/repl_probe.py 2>/dev/null | grep -v amdgpu.ids > $out/prof_${tag}_host_copy_probe.txt
# the run kernels over single runs (64 MiB, 4 GiB) and over 1 GiB with a break every ~33 ... 30 000 bytes: whole calls
{ echo "# tools/probes/run_time.py + tools/probes/run_density.py (MI355X): the run kernels (run_scan.hip), wall time of whole calls, device texts"
  python tools/probes/run_time.py 2>/dev/null | grep -v amdgpu.ids; python tools/probes/run_density.py 1024 2>/dev/null | grep -v amdgpu.ids; } > $out/prof_${tag}_run_probe.txt
bash tools/probes/pair_profile.sh $tag > /dev/null 2>&1
tail -1 $out/prof_${tag}_bench.json | cut -c1-300
head -6 $out/prof_${tag}_kernel_stats.txt | cut -c1-60,91-170
head -14 $out/prof_${tag}_pmc_fetch.txt
