#!/bin/bash
# A/B/C of plane_count builds inside ONE gpurun call (variants: tools/probes/build_variant.sh <suffix> plane_count.hip -D...):
# tools/count_probe.py per variant, the list twice.  usage: count_ab.sh "a b c" [fasta_n]
cd "$GRAFT_REPO_ROOT"
variants=${1:-"a b"}
nf=${2:-50000000}
cp rejit_amd/librejit_hip.so /tmp/librejit_hip_keep.so
for round in 1 2; do
  for v in $variants; do
    cp rejit_amd/librejit_hip_$v.so rejit_amd/librejit_hip.so
    python tools/count_probe.py $nf 200 1 2>/dev/null | grep -v amdgpu | grep "round" | sed "s/^/$v$round /" | cut -c1-160
  done
done
cp /tmp/librejit_hip_keep.so rejit_amd/librejit_hip.so
