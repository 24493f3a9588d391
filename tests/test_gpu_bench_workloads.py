"""GPU tests (-m gpu) of bench.py's workloads as TWO ranks on one GPU (gloo for the collectives,
--same-device): the multi-rank path of every workload the 8-GPU scaling run can use -- shards + halo +
carry exchange (regexdna, literal, complex), file sharding + output gather (jrep) -- gives the result of
the one-rank run.  RCCL itself is exercised only on a multi-GPU node (the driver's SCALE run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--fasta-n", "2000000", "--literal-bytes", "200000000",
         "--tree-files", "300"]


def bench(workload, ranks):
    # `python bench.py --gpus N` as the driver calls it: bench.py starts its own ranks (spawn_ranks)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--workload", workload] + SMALL
    if ranks > 1:
        cmd += ["--backend", "gloo", "--same-device"]
    for attempt in range(2):   # (a rendezvous on a just-freed port can fail once in a while)
        r = subprocess.run(cmd, capture_output=True, timeout=900, cwd=ROOT)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("workload", ["regexdna", "literal", "complex", "jrep"])
def test_two_ranks_equal_one(workload):
    two = bench(workload, 2)
    assert two["n_gpus"] == 2 and two["value"] > 0 and two["scaling"] == "weak" and "roofline" in two
    if workload == "regexdna":
        # weak scaling: two ranks hold a 2 x 20 MB text; one rank over the same 40 MB text must count the same
        one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-extra",
                              "--no-cpu-baseline", "--fasta-n", "4000000"], capture_output=True, timeout=900, cwd=ROOT)
        assert one.returncode == 0, one.stderr.decode()[-2000:]
        ref = json.loads([l for l in one.stdout.decode().splitlines() if l.startswith("{")][-1])
        assert two["matches_per_pass"] == ref["matches_per_pass"]
    elif workload in ("literal", "complex"):
        assert two["matches"] >= 400          # 200 planted per rank, some across the cut; checked one by one inside bench.py
    else:
        assert two["files_with_matches"] >= 1 and two["output_lines"] >= two["files_with_matches"]
