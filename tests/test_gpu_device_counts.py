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


def test_gather_spans_one_rank_rccl_communicator():
    """rj_scan_gather_spans over RCCL itself (ncclAllGather for the carry rows, ncclSend / ncclRecv in one group for the
    pairs; one rank on this box: root sends to itself): the whole list, global offsets, equals the oracle."""
    torch, rejit_amd, o = _setup()
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    rccl = ctypes.CDLL("librccl.so.1")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid, comm = UniqueId(), ctypes.c_void_p()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    ctypes.CDLL(None).fflush(None)
    try:
        rng = random.Random(4)
        rx = b"([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)"
        parts = []
        for _ in range(300):
            parts.append(bytes(rng.choices(b"abcdefghijklmnopqrstuvwxyz 0123", k=rng.randint(100, 3000))))
            parts.append(rng.choice([b"complexregexpabcdefghthe", b"regexpregexpabcdefghat", b"xabcdefghthe", b"ccabcdefghdas well"]))
        text = b"".join(parts)
        t = torch.frombuffer(bytearray(text + bytes(16)), dtype=torch.uint8).to(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        scan = rejit_amd.Scan(rejit_amd.Program(rx))
        want = o.match_all(rx, text)
        for _ in range(2):
            total, spans = scan.gather_spans(t.data_ptr(), len(text), 0, 0, 1, root=0, comm=comm.value, stream=st)
            assert total == len(want) and spans == want
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
        ctypes.CDLL(None).fflush(None)


@pytest.mark.parametrize("world,root", [(2, 0), (3, 2), (5, 1)])
def test_gather_spans_of_shards_on_one_device(world, root):
    """rj_scan_gather_spans_via: `world` shards of one text on the one device (a thread, a stream and an rj_scan each; the
    collectives played by a barrier and device-to-device copies) -- the list on root equals the oracle's over the WHOLE
    text for patterns whose selection travels over the cuts (runs of `a` across every cut), a nullable pattern (the empty
    match at a shard's first position), a bounded and a dense one; root != 0 as well."""
    torch, rejit_amd, o = _setup()
    from rejit_amd import api, sharding
    dev = torch.device("cuda:0")
    rng = random.Random(10 + world)
    parts = []
    while sum(map(len, parts)) < 150_000:
        parts.append(b"a" * rng.choice([1, 2, 3, 5, 4000, 9001]) + bytes(rng.choices(b"bcx\n", k=rng.randint(1, 8))))
    text = b"".join(parts)
    text = text[:70_000] + b"a" * 50_001 + text[70_000:]
    patterns = [(b"aaa", 2), (b"a{2,5}", 4), (b"x*", 0), (b"ab|ba", 1), (b"[bc]+x", None)]
    n = len(text)
    ranges = sharding.partition(n, world, align=1024)
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    barrier = threading.Barrier(world)
    sends = [None] * world
    results, errors = {}, []

    def rank_main(rank):
        try:
            stream = torch.cuda.Stream(dev)

            def allgather(ctx, send, recv, nbytes, st):
                if hip.hipStreamSynchronize(st) != 0:
                    return 1
                sends[rank] = send
                barrier.wait(timeout=120)
                bad = 0
                for r in range(world):
                    bad |= hip.hipMemcpyAsync(recv + r * nbytes, sends[r], nbytes, 3, st)
                bad |= hip.hipStreamSynchronize(st)
                barrier.wait(timeout=120)
                return bad

            def gatherv(ctx, send, send_bytes, recv, offs, sizes, rt, st):
                if hip.hipStreamSynchronize(st) != 0:
                    return 1
                sends[rank] = send
                barrier.wait(timeout=120)
                bad = 0
                if rank == rt:
                    for r in range(world):
                        if sizes[r]:
                            bad |= hip.hipMemcpyAsync(recv + offs[r], sends[r], sizes[r], 3, st)
                    bad |= hip.hipStreamSynchronize(st)
                barrier.wait(timeout=120)
                return bad

            ag, gv = api.ALLGATHER_FN(allgather), api.GATHERV_FN(gatherv)
            for rx, halo in patterns:
                own = ranges[rank]
                # (unbounded patterns: the shard sees the text to its end)
                lo, hi = sharding.visible_range(n, own, halo) if halo is not None else (max(0, (own[0] - 64) & ~15), n)
                shard = torch.frombuffer(bytearray(text[lo:hi] + bytes(16)), dtype=torch.uint8).to(dev)
                scan = rejit_amd.Scan(rejit_amd.Program(rx))
                for _ in range(2):
                    total, spans = scan.gather_spans(shard.data_ptr(), hi - lo, lo, rank, world, root=root, allgather=ag, gatherv=gv,
                                                     own_begin=own[0] - lo, own_end=min(own[1], n + 1) - lo, stream=stream.cuda_stream)
                results[(rank, rx)] = (total, spans)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for rx, _ in patterns:
        want = o.match_all(rx, text)
        for rank in range(world):
            total, spans = results[(rank, rx)]
            assert total == len(want), (rx, rank, total, len(want))
            assert (spans == want) if rank == root else (spans is None), (rx, rank)


def _visible_devices():
    import torch
    return torch.cuda.device_count()


def test_native_samples_over_every_gpu_of_the_node(tmp_path, capsys):
    """The two native multi-GPU callers on min(visible devices, 8) devices of THIS box, each against the oracle:
      * samples/regexdna_rccl.cc -- the nine counts, a thread + shard + RCCL rank per device, rj_multi_device_counts;
      * samples/complex_rccl.cc  -- BASELINE configs[3]'s regex over a synthetic text, the match LIST gathered on device 0
        (rj_scan_gather_spans: ncclAllGather for the carry rows, ncclSend / ncclRecv for the pairs).
    The device count used is printed (pytest -s / the captured output of a failure) and asserted: on a box with several
    GPUs the N >= 2 runs must equal the one-device run and the oracle; on a one-GPU box those assertions are reported as
    skipped, visibly, and the one-rank communicator paths are what is checked."""
    import os
    import subprocess
    import rejit_amd
    from rejit_amd import workloads as W
    from checkers import Oracle
    rejit_amd.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "samples"), "rccl"], stdout=subprocess.DEVNULL)
    visible = _visible_devices()
    use = min(visible, 8)
    o = Oracle()
    # ---- regexdna over `use` devices (the sample falls back to one device below 1 MB per shard: 3 M lines = 30 MB)
    nf = 3_000_000 if use > 1 else 300_000
    raw = W.fasta_raw_numpy(nf).tobytes()
    seq = W.fasta_stripped_numpy(nf).tobytes()
    want_counts = None
    for devices in sorted({1, use}):
        r = subprocess.run([os.path.join(root, "samples", "regexdna_rccl"), "--devices", str(devices)], input=raw, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert ("regexdna_rccl: %d device(s) of %d visible" % (devices, visible)) in r.stderr.decode(), r.stderr.decode()[-500:]
        lines = r.stdout.decode().splitlines()
        pats = [rx for rx in lines if "|" in rx]
        assert len(pats) == 9
        if want_counts is None:
            want_counts = {}
            for line in pats:
                rx, cnt = line.rsplit(" ", 1)
                want_counts[rx] = len(o.match_all(rx.encode(), seq))
        for line in pats:
            rx, cnt = line.rsplit(" ", 1)
            assert int(cnt) == want_counts[rx], (devices, line)
        assert lines[-2:] == [str(len(raw)), str(len(seq))]
    # ---- the complex regex: the list on device 0, one device vs `use` devices vs the oracle
    rx = "([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)"
    nbytes = 24 << 20
    dump = str(tmp_path / "text.bin")
    outs = {}
    for devices in sorted({1, use}):
        cmd = [os.path.join(root, "samples", "complex_rccl"), "--devices", str(devices), "--bytes", str(nbytes), "--plant", "500", "--print-spans"]
        if devices == 1:
            cmd += ["--dump-text", dump]
        r = subprocess.run(cmd, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        lines = r.stdout.decode().splitlines()
        while lines and not lines[0].startswith("devices "):   # (RCCL prints its version banner on stdout)
            lines.pop(0)
        assert lines and lines[0] == "devices %d (visible %d)" % (devices, visible), lines[:2]
        n_matches = int(lines[2].split()[1])
        spans = [tuple(int(v) for v in l.split()) for l in lines[5:]]
        assert len(spans) == n_matches
        outs[devices] = (lines[3], spans)
    with open(dump, "rb") as fh:
        text = fh.read()
    assert len(text) == nbytes
    want = o.match_all(rx.encode(), text)
    assert len(want) >= 500
    for devices, (digest, spans) in outs.items():
        assert spans == want, (devices, len(spans), len(want))
    with capsys.disabled():
        print("\n[multi-GPU] visible devices: %d; samples ran on 1 and on %d device(s); %s" %
              (visible, use, "N >= 2 assertions made" if use >= 2 else "N >= 2 assertions SKIPPED (one GPU on this box)"))
