cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for g in 0 4096 8192 12288 24576 32768; do
  RJ_SCAN_GRID=$g DENSE_PROBE_RX="regexp" python tools/dense_probe.py 5e9 2>/dev/null | grep -v amdgpu | sed "s/^/grid=$g r$round /" | cut -c1-150
done
for ch in 64 96 128 160 240; do
  RJ_PLANE_CHUNKS=$ch python tools/count_probe.py 50000000 200 1 2>/dev/null | grep "how=1" | sed "s/^/plane_chunks=$ch r$round /" | cut -c1-150
done
done
