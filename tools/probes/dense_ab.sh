#!/bin/bash
# A/B/C of dense_streams builds inside ONE gpurun call: the library is swapped between runs of bench.py's dense extras
# (rejit_amd/librejit_hip_{a,b,c}.so: built by hand from variants of dense_streams.hip; scratch copy on the GPU box).
cd "$GRAFT_REPO_ROOT"
cp rejit_amd/librejit_hip.so rejit_amd/librejit_hip_a.so
for round in 1 2; do
  for v in a b c; do
    cp rejit_amd/librejit_hip_$v.so rejit_amd/librejit_hip.so
    timeout 120 python bench.py --no-big --no-cpu-baseline > gpurun_out/ab_$v$round.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('gpurun_out/ab_$v$round.json').read().strip().splitlines()[-1])
print('$v$round', ' '.join('%s %.4f %.4f' % (k, d[k]['roofline']['avg_launch_ms'], d[k]['roofline']['frac']) for k in ('dense_scan','dense_select','line_table')))
PY
  done
done
cp rejit_amd/librejit_hip_a.so rejit_amd/librejit_hip.so
