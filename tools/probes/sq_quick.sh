root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sq
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/prof_sq -o r -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-big > /dev/null 2> /tmp/sq.log
for c in SQ_WAVES SQ_INSTS_VALU; do echo "## $c"; python $root/tools/pmc_summary.py /tmp/prof_sq $c | head -14; done > $root/gpurun_out/sq_quick.txt
tail -3 /tmp/sq.log
