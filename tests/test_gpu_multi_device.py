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
# (the last three are at risk of the reference's ring artefact: sharded by segment ownership, every shard sees the text to its end)
for rx in [b"regexp", b"aa", b"ab|bcx", b"a{1,3}", b"^a", b"b$", b"(ab|ba)x?", b"[ab]{2,5}c", b"x", b"$", b".{0,2}.", b"[ab]{1,3}[ab]", b"(a|ab)(c|bcx)"]:
    p = rejit_amd.Program(rx)
    out[rx.decode()] = p.match_all(text)
# Where the synchronisation points of the reference's loop lie depends on what is alive (ADVICE r04): a run of [ab] that
# begins a few bytes before a shard's buffer would begin (cuts are multiples of 4096; round 4 gave a shard its text from
# cut - 64 on) and crosses the cut makes `[ab]{1,3}[ab]` match every 4 bytes IN PHASE WITH THE RUN'S BEGIN; a shard that
# took its buffer's first byte for a point found the phase of ITS buffer, and the two neighbours disagreed on the segment
# at the cut (matches twice, or shifted).  Also: a point exactly at a cut with a match beginning there.
seg = bytearray(rng.choices(b"abx", k=1_200_000))
for k, cut in enumerate(range(4096, len(seg) - 4096, 4096)):
    if k %% 2:
        begin = cut - 64 - rng.choice([1, 2, 3, 5, 6, 7, 9, 30, 31, 33])
        seg[begin - 1] = ord("x")
        seg[begin:cut + 300] = bytes(rng.choices(b"ab", k=cut + 300 - begin))
    else:
        seg[cut - 1] = ord("x")
        seg[cut:cut + 3] = rng.choice([b"aba", b"abb", b"bab"])
seg = bytes(seg)
for rx in [b"[ab]{1,3}[ab]", b".{0,2}.", b"(ab|ba)+", b"[ab]{2}[ab]?"]:
    out["cut:" + rx.decode()] = rejit_amd.Program(rx).match_all(seg)
files = [bytes(rng.choices(b"ab regexp\n", k=rng.choice([0, 5, 300, 4000, 70000]))) for _ in range(200)]
p = rejit_amd.Program(b"regexp|^a")
out["batch"] = p.match_all_batch(files)
out["risk"] = [rejit_amd.Program(rx).info()["ring_artefact_risk"] for rx in (b".{0,2}.", b"[ab]{1,3}[ab]", b"(a|ab)(c|bcx)")]
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
    import json
    assert any(json.loads(one)["risk"]), "none of the at-risk patterns is flagged: the segment-ownership split went untested"


def test_carry_decide_kernel_equals_the_host_protocol():
    """rj_carry_decide (the device half of sharding.CarryExchange) against sharding.must_rerun on random rows."""
    import random
    import torch
    import rejit_amd
    from rejit_amd import api, sharding
    rejit_amd.build()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    rng = random.Random(5)
    for trial in range(200):
        world, P = rng.randint(1, 8), rng.randint(1, 9)
        rows = torch.full((world, P, 8), -1, dtype=torch.int64)
        pos = 0
        for r in range(world):
            for p in range(P):
                k = rng.choice([0, 0, 1, 3])
                rows[r, p, 0] = k
                if k:
                    fb = 1000 * r + rng.randint(0, 30)
                    fe = fb + rng.choice([0, 1, 5])
                    lb = max(fb, 1000 * r + rng.randint(0, 999))
                    le = lb + rng.choice([0, 1, 5, 40])
                    rows[r, p, 1:5] = torch.tensor([fb, fe, lb, le])
                used = rng.choice([(0, 0, 0), (0, 0, 0), (1000 * r - rng.randint(0, 3) + 2, 1000 * r - 1, 1)])
                rows[r, p, 5:8] = torch.tensor(used)
        rank = rng.randrange(world)
        out = torch.zeros(4 * P + 1, dtype=torch.int64).pin_memory()
        api.carry_decide(rows.to(dev), world, rank, P, out, st)
        torch.cuda.synchronize(dev)
        out = out.tolist()
        again = 0
        for p in range(P):
            assert out[p] == int(rows[:, p, 0].sum())
            for r in range(1, world):
                prev = [q for q in range(r) if rows[q, p, 3] >= 0]
                cin = (0, 0, False)
                if prev:
                    lb, le = int(rows[prev[-1], p, 3]), int(rows[prev[-1], p, 4])
                    cin = (le if le > lb else lb + 1, le, True)
                row = rows[r, p].tolist()
                fe = (row[1], row[2]) if row[1] >= 0 else None
                need = sharding.must_rerun(fe, (row[5], row[6], bool(row[7])), cin)
                again |= int(need)
                if r == rank:
                    assert out[P + p] == int(need), (trial, r, p)
                    if need:
                        assert (out[2 * P + 2 * p], out[2 * P + 2 * p + 1]) == (cin[0], cin[1])
        assert out[4 * P] == again
