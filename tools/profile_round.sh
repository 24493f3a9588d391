#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats of the default bench command and the two PMC
# passes (own runs, --kernel-trace only).  Summaries land in gpurun_out/prof_<tag>_*.txt
tag=${1:-r01}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python $root/bench.py > $out/prof_${tag}_bench.json 2> /tmp/kt.log
python $root/tools/prof_summary.py $(find /tmp/prof_kt -name "*.db" | head -1) > $out/prof_${tag}_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/f.log
python $root/tools/pmc_summary.py /tmp/prof_f FETCH_SIZE > $out/prof_${tag}_pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/w.log
python $root/tools/pmc_summary.py /tmp/prof_w WRITE_SIZE > $out/prof_${tag}_pmc_write.txt
tail -1 $out/prof_${tag}_bench.json | cut -c1-400
head -14 $out/prof_${tag}_pmc_fetch.txt
