"""GPU tests (-m gpu) of the multi-device paths BELOW the C boundary (rejit_amd/csrc/multi_device.hip): a
rejit.h / C-ABI caller's single rj_match_all / rj_match_all_batch call spread over every visible GPU.
On a one-GPU box the shards are virtual (RJ_VIRTUAL_DEVICES: k shards on the one device, each with its own
worker thread, scan object and stream), which exercises everything but the second PCIe link: the
partition + halo, the carry of the selection over the cuts, the re-run rule, the packing of files onto
shards and the reassembly in the caller's order -- results must equal the one-device call bit for bit."""
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import random, sys, json
sys.path.insert(0, %(root)r)
import rejit_amd
rng = random.Random(5)
out = {}
text = bytes(rng.choices(b"aabbcx\n regexp", k=3_000_000)) + b"aaaa" * 1000
for rx in [b"regexp", b"aa", b"ab|bcx", b"a{1,3}", b"^a", b"b$", b"(ab|ba)x?", b"[ab]{2,5}c", b"x", b"$"]:
    p = rejit_amd.Program(rx)
    out[rx.decode()] = p.match_all(text)
files = [bytes(rng.choices(b"ab regexp\n", k=rng.choice([0, 5, 300, 4000, 70000]))) for _ in range(200)]
p = rejit_amd.Program(b"regexp|^a")
out["batch"] = p.match_all_batch(files)
print(json.dumps(out))
"""


def run(virtual):
    env = dict(os.environ)
    env.pop("RJ_VIRTUAL_DEVICES", None)
    if virtual:
        env["RJ_VIRTUAL_DEVICES"] = str(virtual)
        env["RJ_MULTI_DEVICE_MIN_BYTES"] = "100000"
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return r.stdout.decode().strip().splitlines()[-1]


@pytest.mark.parametrize("virtual", [2, 3, 7])
def test_shards_equal_one_device(virtual):
    one = run(0)
    assert run(virtual) == one
