cd /tmp && export TMPDIR=/tmp
export DENSE_PROBE_RX="[a-f]+[0-9]"
for v in run generic; do
  rm -rf /tmp/pmc_$v
  if [ $v = generic ]; then export RJ_NO_RUN_STEPS=1; else unset RJ_NO_RUN_STEPS; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc_$v -o r -- python $GRAFT_REPO_ROOT/tools/dense_probe.py 5e9 4 > /dev/null 2>&1
  for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU; do echo "$v $c: $(python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_$v $c | grep dense_streams | cut -c1-110)"; done
done
