"""GPU tests (-m gpu) of rj_multi_device_counts (include/rejit_hip.h): the sharded multi-pattern count with its carry
exchange behind ONE C call, for C++ callers with a process (or thread) per GPU.

* over RCCL: a one-rank communicator made here with ncclCommInitRank (the box has one GPU) -- the library binds
  ncclAllGather with dlopen and must find the copy the process has loaded;
* over a caller-supplied all-gather (rj_multi_device_counts_via): three shards of one text on the one device, a thread,
  a stream and an rj_multi per shard, the collective played by a barrier and device-to-device copies.  The texts are
  chosen so that the selection DOES travel over the cuts (runs of `a` under `aaa`, every cut inside a run): the counts
  must be those of the whole text on every rank."""
import ctypes
import random
import threading

import pytest

pytestmark = pytest.mark.gpu

DNA = [b"agggtaaa|tttaccct", b"[cgt]gggtaaa|tttaccc[acg]", b"a[act]ggtaaa|tttacc[agt]t", b"ag[act]gtaaa|tttac[agt]ct"]


def _setup():
    import torch
    import rejit_amd
    from checkers import Oracle
    rejit_amd.build()
    return torch, rejit_amd, Oracle()


def test_one_rank_rccl_communicator():
    torch, rejit_amd, o = _setup()
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    rccl = ctypes.CDLL("librccl.so.1")        # (torch has loaded it: the same copy)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid, comm = UniqueId(), ctypes.c_void_p()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    ctypes.CDLL(None).fflush(None)   # (RCCL's version banner sits in C stdio's buffer: out now, inside this test's capture, not at exit)
    try:
        rng = random.Random(3)
        text = bytes(rng.choices(b"acgt", k=400_000)) + b"agggtaaa" * 3 + b"tttaccct"
        t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
        progs = [rejit_amd.Program(p) for p in DNA]
        multi = rejit_amd.MultiScan(progs)
        st = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(2):
            got = multi.device_counts(t.data_ptr(), len(text), 0, 0, 1, comm=comm.value, stream=st)
            assert got == [len(o.match_all(p, text)) for p in DNA]
        assert multi.scan(0).spans() == o.match_all(DNA[0], text)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
        ctypes.CDLL(None).fflush(None)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_shards_on_one_device_over_a_callers_all_gather(world):
    torch, rejit_amd, o = _setup()
    from rejit_amd import api, sharding
    dev = torch.device("cuda:0")
    rng = random.Random(world)
    # runs of `a` long enough to cross every cut; lengths such that the selection under the true carry differs from
    # the one under the empty carry (a shard that starts inside a run)
    parts = []
    while sum(map(len, parts)) < 200_000:
        parts.append(b"a" * rng.choice([1, 2, 3, 5, 4000, 9001]) + bytes(rng.choices(b"bcx\n", k=rng.randint(1, 8))))
    text = b"".join(parts)
    text = text[:100_000] + b"a" * 60_001 + text[100_000:]
    patterns = [b"aaa", b"aa|b", b"a{2,5}", b"ab|ba", b"x"]
    want = [len(o.match_all(p, text)) for p in patterns]
    n = len(text)
    ranges = sharding.partition(n, world, align=1024)
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    barrier = threading.Barrier(world)
    sends = [None] * world
    results, errors = [None] * world, []

    def rank_main(rank):
        try:
            own = ranges[rank]
            lo, hi = sharding.visible_range(n, own, 5)
            shard = torch.frombuffer(bytearray(text[lo:hi]), dtype=torch.uint8).to(dev)
            stream = torch.cuda.Stream(dev)
            multi = rejit_amd.MultiScan([rejit_amd.Program(p) for p in patterns])

            def allgather(ctx, send, recv, nbytes, st):
                if hip.hipStreamSynchronize(st) != 0:
                    return 1
                sends[rank] = send
                barrier.wait(timeout=120)
                bad = 0
                for r in range(world):
                    bad |= hip.hipMemcpyAsync(recv + r * nbytes, sends[r], nbytes, 3, st)   # hipMemcpyDeviceToDevice
                bad |= hip.hipStreamSynchronize(st)     # (nobody rewrites its rows before every copy of them is done)
                barrier.wait(timeout=120)
                return bad

            cb = api.ALLGATHER_FN(allgather)
            for _ in range(2):
                results[rank] = multi.device_counts(shard.data_ptr(), hi - lo, lo, rank, world, allgather=cb, own_begin=own[0] - lo,
                                                    own_end=own[1] - lo, stream=stream.cuda_stream)
            spans = [[(b + lo, e + lo) for b, e in multi.scan(i).spans()] for i in range(len(patterns))]
            results[rank] = (results[rank], spans)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for rank in range(world):
        assert results[rank][0] == want, (rank, results[rank][0], want)
    for i, p in enumerate(patterns):
        joined = [s for rank in range(world) for s in results[rank][1][i]]
        assert joined == o.match_all(p, text), p


def test_native_sample_over_every_gpu_of_the_node(tmp_path):
    """samples/regexdna_rccl.cc: one process, a thread + shard + RCCL rank per visible device (here: one), the exchange
    behind rj_multi_device_counts -- its nine count lines against the oracle over the stripped sequence."""
    import os
    import subprocess
    import rejit_amd
    from rejit_amd import workloads as W
    from checkers import Oracle
    rejit_amd.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "samples"), "rccl"], stdout=subprocess.DEVNULL)
    nf = 300000
    raw = W.fasta_raw_numpy(nf).tobytes()
    seq = W.fasta_stripped_numpy(nf).tobytes()
    r = subprocess.run([os.path.join(root, "samples", "regexdna_rccl")], input=raw, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = r.stdout.decode().splitlines()
    o = Oracle()
    pats = [rx for rx in lines if "|" in rx]
    assert len(pats) == 9
    for line in pats:
        rx, cnt = line.rsplit(" ", 1)
        assert int(cnt) == len(o.match_all(rx.encode(), seq)), line
    assert lines[-2:] == [str(len(raw)), str(len(seq))]
