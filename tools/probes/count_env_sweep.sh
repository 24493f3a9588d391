#!/bin/bash
# tools/count_probe.py (counts mode) under environment overrides, the list twice, inside ONE gpurun call.
# usage: count_env_sweep.sh "X=1" "RJ_COUNT_LDS_PAD=12000" ...
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for env in "$@"; do
  env $env python tools/count_probe.py 50000000 200 1 2>/dev/null | grep "counts_only=1" | sed "s/^/$env r$round /" | cut -c1-170
done
done
