#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the bench line, the kernel-trace stats of the headline run and
# the two PMC passes (own runs, --kernel-trace only).  Summaries land in gpurun_out/prof_<tag>_*
tag=${1:-r01}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
python $root/bench.py > $out/prof_${tag}_bench.json 2> /tmp/b.log
# headline run only (no extras, no CPU sample): the per-kernel averages then are those of the timed region
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python $root/bench.py --no-extra --no-cpu-baseline > $out/prof_${tag}_bench_profiled.json 2> /tmp/kt.log
python $root/tools/prof_summary.py $(find /tmp/prof_kt -name "*.db" | head -1) > $out/prof_${tag}_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/f.log
python $root/tools/pmc_summary.py /tmp/prof_f FETCH_SIZE > $out/prof_${tag}_pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/w.log
python $root/tools/pmc_summary.py /tmp/prof_w WRITE_SIZE > $out/prof_${tag}_pmc_write.txt
tail -1 $out/prof_${tag}_bench.json | cut -c1-300
tail -1 $out/prof_${tag}_bench_profiled.json | cut -c1-300
grep -v "at::native" $out/prof_${tag}_kernel_stats.txt | head -8
