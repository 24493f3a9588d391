#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the bench line, kernel-trace stats of the headline run and of the full
# bench, and the PMC passes (own runs, --kernel-trace only -- never combined with other trace domains).
# Summaries land in gpurun_out/prof_<tag>_*; tools/collect_profiles.py turns them into profiles/<tag>_*.
tag=${1:-r02}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_full /tmp/prof_f /tmp/prof_w /tmp/prof_sq /tmp/prof_lin
python $root/bench.py > $out/prof_${tag}_bench.json 2> /tmp/b.log
# headline run only (no extras, no CPU sample): the per-kernel averages then are those of the timed region
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python $root/bench.py --no-extra --no-cpu-baseline > $out/prof_${tag}_bench_profiled.json 2> /tmp/kt.log
python $root/tools/rocpd_stats.py $(find /tmp/prof_kt -name "*.db" | head -1) 12 > $out/prof_${tag}_kernel_stats.txt
# every extra of the bench line (fused, literal, complex, behind, dense, line table): one table
rocprofv3 --kernel-trace --stats -d /tmp/prof_full -o r -- python $root/bench.py --no-cpu-baseline --no-big --steps 5 > /dev/null 2> /tmp/full.log
python $root/tools/rocpd_stats.py $(find /tmp/prof_full -name "*.db" | head -1) 40 > $out/prof_${tag}_kernel_stats_full.txt
# the linear-time carry scan and the dense / class patterns
rocprofv3 --kernel-trace --stats -d /tmp/prof_lin -o r -- python $root/tools/linear_probe.py > $out/prof_${tag}_linear_probe.txt 2> /tmp/lin.log
python $root/tools/rocpd_stats.py $(find /tmp/prof_lin -name "*.db" | head -1) 24 > $out/prof_${tag}_kernel_stats_linear.txt
# PMC passes: one counter group per run
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-big > /dev/null 2> /tmp/f.log
python $root/tools/pmc_summary.py /tmp/prof_f FETCH_SIZE > $out/prof_${tag}_pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-big > /dev/null 2> /tmp/w.log
python $root/tools/pmc_summary.py /tmp/prof_w WRITE_SIZE > $out/prof_${tag}_pmc_write.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_sq -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-big > /dev/null 2> /tmp/sq.log
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY; do
  echo "## $c"; python $root/tools/pmc_summary.py /tmp/prof_sq $c | head -14
done > $out/prof_${tag}_pmc_sq.txt
tail -3 /tmp/sq.log | cut -c1-300
tail -1 $out/prof_${tag}_bench.json | cut -c1-400
head -8 $out/prof_${tag}_kernel_stats.txt | cut -c1-60,91-170
head -12 $out/prof_${tag}_pmc_fetch.txt
head -30 $out/prof_${tag}_pmc_sq.txt
