#!/bin/bash
# Every GPU fuzzer once with a fresh seed (tools/fuzz_sweep.sh <seed> [seconds per fuzzer]): the last lines of each under gpurun_out/fuzz_sweep_<seed>.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
seed=${1:-1}; lim=${2:-150}
out=gpurun_out/fuzz_sweep_$seed.txt; : > $out
run() { echo "== $*" >> $out; timeout $lim "$@" > $out.one 2>&1; rc=$?; tail -4 $out.one >> $out; [ $rc -eq 124 ] && echo "TIMED OUT after $lim s (no summary)" >> $out; echo "rc=$rc" >> $out; rm -f $out.one; }   # (VAR=x run ...: the environment reaches the command)
run python tools/fuzz_counts.py 300 $((seed + 100))
run python tools/fuzz_counts_general.py 400 $((seed + 200))
run python tools/fuzz_runs.py 1500 $((seed + 300))
run python tools/fuzz_run_steps.py 400 $((seed + 400))
run python tools/fuzz_select_gpu.py 600 $((seed + 500))
run python tools/fuzz_more.py 300 $((seed + 600))
run python tools/fuzz_large.py 1500 $((seed + 700))
run python tools/fuzz_reuse.py 40 $((seed + 800))
run python tools/fuzz_churn.py 100 $((seed + 900)) 3
run python tools/fuzz_verify_lds.py 800 $((seed + 1000))
run python tools/fuzz_fused.py $((seed + 1100))
run python tools/fuzz_floating.py $((seed + 1200))
RJ_NO_SMALL=1 run python tools/fuzz_pairs.py 400 $((seed + 1300))
cat $out
