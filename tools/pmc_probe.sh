#!/bin/bash
# Run ON THE GPU BOX: PMC passes (own runs, --kernel-trace only) over a probe script.
#   tools/pmc_probe.sh <out-prefix> <python script + args...>
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out; mkdir -p $out
pre=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_sq -o r -- python "$@" > /tmp/pmc_sq.log 2>&1
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY; do
  echo "## $c"; python $root/tools/pmc_summary.py /tmp/pmc_sq $c | head -12
done > $out/${pre}_pmc_sq.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o r -- python "$@" > /tmp/pmc_f.log 2>&1
python $root/tools/pmc_summary.py /tmp/pmc_f FETCH_SIZE | head -12 > $out/${pre}_pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o r -- python "$@" > /tmp/pmc_w.log 2>&1
python $root/tools/pmc_summary.py /tmp/pmc_w WRITE_SIZE | head -12 > $out/${pre}_pmc_write.txt
