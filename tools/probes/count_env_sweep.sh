cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for env in "X=1" "RJ_COUNT_BATCH=32" "RJ_COUNT_BATCH=16" "RJ_COUNT_BATCH=8" "RJ_COUNT_CHUNKS=120" "RJ_COUNT_CHUNKS=240" "RJ_COUNT_CHUNKS=320" "RJ_COUNT_CHUNKS=480"; do
  env $env python tools/count_probe.py 50000000 200 1 2>/dev/null | grep "counts_only=1" | sed "s/^/$env r$round /" | cut -c1-170
done
done
