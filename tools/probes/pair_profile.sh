#!/bin/bash
# ON THE GPU BOX: the pair kernels (run_scan.hip: `"[^"]*"`) -- whole calls with and without them, per-kernel times, FETCH_SIZE / WRITE_SIZE
# (PMC passes on their own, --kernel-trace only).  bash tools/probes/pair_profile.sh <tag> -> gpurun_out/prof_<tag>_pair_*.txt
tag=${1:-r06}
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp_kt /tmp/pp_f /tmp/pp_w
{ echo "# tools/probes/pair_time.py 1024 (MI355X): whole calls over 1 GiB device texts; then the same with RJ_NO_PAIRS=1 (the paths before: windows + walks, carry scan)"
  python $root/tools/probes/pair_time.py 1024 2>/dev/null | grep -v amdgpu.ids
  echo "# RJ_NO_PAIRS=1"
  RJ_NO_PAIRS=1 python $root/tools/probes/pair_time.py 1024 2>/dev/null | grep -v amdgpu.ids; } > $out/prof_${tag}_pair_probe.txt
rocprofv3 --kernel-trace --stats -d /tmp/pp_kt -o r -- python $root/tools/probes/pair_time.py 1024 > /dev/null 2> /tmp/pp_kt.log
python $root/tools/rocpd_stats.py $(find /tmp/pp_kt -name "*.db" | head -1) 12 > $out/prof_${tag}_pair_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pp_f -o r -- python $root/tools/probes/pair_time.py 1024 > /dev/null 2> /tmp/pp_f.log
python $root/tools/pmc_summary.py /tmp/pp_f FETCH_SIZE | head -12 > $out/prof_${tag}_pair_pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pp_w -o r -- python $root/tools/probes/pair_time.py 1024 > /dev/null 2> /tmp/pp_w.log
python $root/tools/pmc_summary.py /tmp/pp_w WRITE_SIZE | head -12 > $out/prof_${tag}_pair_pmc_write.txt
cat $out/prof_${tag}_pair_probe.txt; head -12 $out/prof_${tag}_pair_kernel_stats.txt | cut -c1-200; cat $out/prof_${tag}_pair_pmc_fetch.txt | cut -c1-200
