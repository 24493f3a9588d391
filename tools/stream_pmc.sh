#!/bin/bash
# GPU box: SQ counters of the bit-stream dense kernel over 1 GB of random ASCII (own PMC pass, --kernel-trace only)
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ds
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_ds -o r -- python $root/tools/dense_probe.py ${1:-1e9} 3 > /tmp/ds.log 2>&1
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY; do
  echo "## $c"; python $root/tools/pmc_summary.py /tmp/prof_ds $c | grep -i "dense_streams\|kernel " | cut -c1-200
done > $out/${2:-r04_stream_pmc}.txt
tail -3 /tmp/ds.log
