"""ctypes binding of include/rejit_hip.h + the in-tree build of librejit_hip.so."""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Tuple

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
SOURCES = ["kernels.hip", "scan_windows.hip", "dense_walk.hip", "select_kernels.hip", "plane_scan.hip", "plane_count.hip", "run_scan.hip", "emit_scan.hip", "dense_streams.hip", "dense_streams_select.hip", "verify_lds.hip", "carry_kernels.hip", "engine.hip", "multi_pattern.hip", "host_api.hip", "linear.hip", "exact_replay.hip", "multi_device.hip", "parser.cc", "lowering.cc", "rejit_api.cc"]
HEADERS = ["kernels.h", "device_program.h", "lowering.h", "carry_scan.h", "behind_walk.h", "exact_replay.h", "engine_internal.h", "table_layout.h", "lds_walk.h", "trace_stamp.h", "dense_swar.h", "dense_streams.h", "tile_lookback.h", "exact_count.h", "short_walk.h", "run_scan.h", "kernel_util.h", "stream_load.h", "dense_streams.hip"]  # (dense_streams_select.hip includes dense_streams.hip)
LIB = os.path.join(PKG, "librejit_hip.so")

_u64p = ctypes.POINTER(ctypes.c_uint64)


class RejitError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"rejit_hip status {status}: {message}")
        self.status = status
        self.message = message


def library_path() -> str:
    return LIB


HASH = os.path.join(PKG, "librejit_hip.srchash")


def _source_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(PKG, "..", "include", f) for f in ("rejit.h", "rejit_hip.h")]
    deps += [os.path.join(PKG, "..", "tools", "probes", "read_probe.hip")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale() -> bool:
    """The library is rebuilt when the CONTENT of a source differs from what it was built
    from (mtimes do not survive the copy to the GPU box)."""
    if not os.path.exists(LIB) or not os.path.exists(HASH) or not os.path.exists(os.path.join(PKG, "librejit_bench.so")):
        return True
    try:
        with open(HASH) as fh:
            return fh.read().strip() != _source_hash()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP/C++ source for gfx950 into rejit_amd/librejit_hip.so (in-tree, so
    the built library travels with the repo snapshot).  hipcc cross-compiles without a GPU."""
    if not force and not _stale():
        return LIB
    # several ranks of one job may arrive here together: serialise, and re-check under the lock
    import fcntl
    lock = open(os.path.join(PKG, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not _stale():
            return LIB
        return _build_locked(verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    # one object per source, compiled side by side (kernels.hip alone takes about a minute), then linked
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread"]

    # an object is kept when it was compiled from this very source and the very headers IT includes (content hash beside
    # it): kernels.hip alone takes about a minute, a change to one file -- or to a header three sources use -- should cost
    # those files only
    import hashlib
    import re
    inc_re = re.compile(rb'^\s*#\s*include\s+"([^"]+)"', re.M)

    def closure(path, seen):
        path = os.path.normpath(path)
        if path in seen or not os.path.exists(path):
            return
        seen.add(path)
        with open(path, "rb") as fh:
            body = fh.read()
        for m in inc_re.finditer(body):
            closure(os.path.join(os.path.dirname(path), m.group(1).decode()), seen)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        deps = set()
        closure(os.path.join(CSRC, src), deps)
        hh = hashlib.sha256(" ".join(flags).encode())
        for d in sorted(deps):
            hh.update(os.path.basename(d).encode())
            with open(d, "rb") as fh:
                hh.update(fh.read())
        want = hh.hexdigest()
        stamp = obj + ".srchash"
        try:
            if os.path.exists(obj) and open(stamp).read().strip() == want:
                return obj
        except OSError:
            pass
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(stamp, "w") as fh:
            fh.write(want + "\n")
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    # the measurement library (bench.py's read-only ceiling): its own shared object, not part of the product's ABI
    probe_src = os.path.join(PKG, "..", "tools", "probes", "read_probe.hip")
    cmd = [hipcc] + flags + ["-shared", probe_src, "-o", BENCH_LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(BENCH_LIB + ".tmp", BENCH_LIB)
    with open(HASH, "w") as f:
        f.write(_source_hash() + "\n")
    return LIB


class _Info(ctypes.Structure):
    _fields_ = [("n_positions", ctypes.c_int32), ("n_words", ctypes.c_int32), ("has_assertions", ctypes.c_int32),
                ("scan_mode", ctypes.c_int32), ("n_windows", ctypes.c_int32), ("window_offset", ctypes.c_uint32),
                ("window_len", ctypes.c_uint32), ("min_len", ctypes.c_uint64), ("max_len", ctypes.c_uint64),
                ("ring_artefact_risk", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class _Stats(ctypes.Structure):
    _fields_ = [("n_hits", ctypes.c_uint64), ("n_candidates", ctypes.c_uint64), ("n_matches", ctypes.c_uint64),
                ("scan_ms", ctypes.c_float), ("total_ms", ctypes.c_float), ("retries", ctypes.c_int32),
                ("large_path", ctypes.c_int32), ("exact_path", ctypes.c_int32), ("linear_path", ctypes.c_int32),
                ("stream_path", ctypes.c_int32), ("slow_starts", ctypes.c_int32), ("count_path", ctypes.c_int32), ("run_path", ctypes.c_int32)]


_lib = None

# every symbol include/rejit_hip.h declares
# rj_allgather_fn (include/rejit_hip.h): ctx, d_send, d_recv, bytes per rank, hip stream -> 0 on success
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p)
# rj_gatherv_fn: ctx, d_send, send_bytes, d_recv (root only), recv_offsets[world], recv_bytes[world], root, hip stream -> 0 on success
GATHERV_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64),
                              ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_void_p)

C_ABI_SYMBOLS = ["rj_compile", "rj_program_free", "rj_program_info", "rj_last_error", "rj_match_full",
                 "rj_match_anywhere", "rj_match_first", "rj_match_all", "rj_free_spans", "rj_scan_create",
                 "rj_scan_destroy", "rj_scan_run", "rj_scan_device_spans", "rj_scan_copy_spans", "rj_scan_stats",
                 "rj_scan_match_full", "rj_device_count", "rj_replace_all", "rj_free_text", "rj_scan_replace",
                 "rj_match_all_batch", "rj_multi_create", "rj_multi_destroy", "rj_multi_run", "rj_multi_scan",
                 "rj_multi_scan_ms", "rj_scan_start", "rj_scan_finish", "rj_multi_set_mode", "rj_multi_run_range",
                 "rj_multi_bounds", "rj_batch_separator", "rj_match_all_packed", "rj_host_alloc", "rj_host_free",
                 "rj_multi_bounds_device", "rj_carry_decide", "rj_multi_start", "rj_multi_finish", "rj_multi_order_after",
                 "rj_multi_device_counts", "rj_multi_device_counts_via", "rj_multi_set_tail_stream", "rj_multi_set_timing", "rj_scan_set_timing", "rj_set_default_timing",
                 "rj_scan_gather_spans", "rj_scan_gather_spans_via", "rj_scan_gathered_spans", "rj_multi_set_counts_only", "rj_scan_stats_sized", "rj_scan_copy_gathered_spans", "rj_scan_count", "rj_host_stats", "rj_replace_all_begin", "rj_replace_all_fetch"]


def load_library():
    """Load the product library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise FileNotFoundError(f"{LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        # PyTorch ships its own copy of the HIP runtime.  If librejit_hip.so (linked against
        # /opt/rocm's) is loaded first, the process ends up with two runtimes and the later one sees
        # no device; loading torch first makes both use the one runtime.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB)
    vp, cp, sz, i64, u64 = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_uint64
    L.rj_compile.restype = ctypes.c_int
    L.rj_compile.argtypes = [cp, ctypes.POINTER(vp)]
    L.rj_program_free.argtypes = [vp]
    L.rj_program_info.argtypes = [vp, ctypes.POINTER(_Info)]
    L.rj_last_error.restype = cp
    L.rj_match_full.argtypes = [vp, cp, sz]
    L.rj_match_anywhere.argtypes = [vp, cp, sz]
    L.rj_match_first.argtypes = [vp, cp, sz, _u64p, _u64p]
    L.rj_match_all.restype = i64
    L.rj_match_all.argtypes = [vp, cp, sz, ctypes.POINTER(_u64p)]
    L.rj_free_spans.argtypes = [_u64p]
    L.rj_match_all_batch.restype = i64
    L.rj_match_all_batch.argtypes = [vp, ctypes.POINTER(cp), ctypes.POINTER(sz), sz, _u64p, ctypes.POINTER(_u64p)]
    L.rj_scan_create.argtypes = [vp, ctypes.POINTER(vp)]
    L.rj_multi_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(vp)]
    L.rj_multi_destroy.argtypes = [vp]
    L.rj_multi_run.argtypes = [vp, vp, u64, _u64p, vp]
    L.rj_multi_run_range.argtypes = [vp, vp, u64, u64, u64, _u64p, vp]
    L.rj_multi_start.argtypes = [vp, vp, u64, u64, u64, vp]
    L.rj_multi_finish.argtypes = [vp, _u64p]
    L.rj_multi_order_after.argtypes = [vp, vp]
    L.rj_multi_scan.restype = vp
    L.rj_multi_scan.argtypes = [vp, ctypes.c_int]
    L.rj_multi_scan_ms.restype = ctypes.c_float
    L.rj_multi_scan_ms.argtypes = [vp]
    L.rj_multi_set_mode.argtypes = [vp, ctypes.c_int]
    L.rj_multi_set_tail_stream.argtypes = [vp, ctypes.c_int]
    L.rj_multi_set_timing.argtypes = [vp, ctypes.c_int]
    L.rj_multi_set_counts_only.argtypes = [vp, ctypes.c_int]
    L.rj_scan_stats_sized.argtypes = [vp, vp, sz]
    L.rj_scan_copy_gathered_spans.restype = i64
    L.rj_scan_copy_gathered_spans.argtypes = [vp, _u64p, u64]
    L.rj_scan_set_timing.argtypes = [vp, ctypes.c_int]
    L.rj_set_default_timing.argtypes = [ctypes.c_int]
    L.rj_set_default_timing(1)   # bench.py, the tests and the tools read scan_ms: the scan kernel's start event is on for them
    L.rj_multi_bounds.argtypes = [vp, _u64p, vp]
    L.rj_multi_bounds_device.argtypes = [vp, ctypes.c_int64, ctypes.c_int, vp, vp]
    L.rj_carry_decide.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    L.rj_multi_device_counts.argtypes = [vp, vp, u64, u64, u64, ctypes.c_int64, vp, ctypes.c_int, ctypes.c_int, _u64p, vp]
    L.rj_multi_device_counts_via.argtypes = [vp, vp, u64, u64, u64, ctypes.c_int64, ALLGATHER_FN, vp, ctypes.c_int, ctypes.c_int, _u64p, vp]
    L.rj_scan_destroy.argtypes = [vp]
    L.rj_scan_gather_spans.restype = i64
    L.rj_scan_gather_spans.argtypes = [vp, vp, u64, u64, u64, ctypes.c_int64, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.rj_scan_gather_spans_via.restype = i64
    L.rj_scan_gather_spans_via.argtypes = [vp, vp, u64, u64, u64, ctypes.c_int64, ALLGATHER_FN, GATHERV_FN, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.rj_scan_gathered_spans.restype = vp
    L.rj_scan_gathered_spans.argtypes = [vp, _u64p]
    L.rj_scan_run.restype = i64
    L.rj_host_stats.restype = ctypes.c_int
    L.rj_host_stats.argtypes = [vp, vp, sz]
    L.rj_scan_count.restype = i64
    L.rj_scan_count.argtypes = [vp, vp, u64, vp]
    L.rj_scan_run.argtypes = [vp, vp, u64, u64, u64, u64, u64, ctypes.c_int, vp]
    L.rj_scan_start.argtypes = [vp, vp, u64, vp]
    L.rj_scan_finish.restype = i64
    L.rj_scan_finish.argtypes = [vp]
    L.rj_scan_device_spans.restype = vp
    L.rj_scan_device_spans.argtypes = [vp]
    L.rj_scan_copy_spans.restype = i64
    L.rj_scan_copy_spans.argtypes = [vp, _u64p, u64]
    L.rj_scan_stats.argtypes = [vp, ctypes.POINTER(_Stats)]
    L.rj_scan_match_full.argtypes = [vp, vp, u64, vp]
    L.rj_replace_all_begin.restype = i64
    L.rj_replace_all_begin.argtypes = [vp, cp, sz, cp, sz, ctypes.POINTER(sz)]
    L.rj_replace_all_fetch.restype = ctypes.c_int
    L.rj_replace_all_fetch.argtypes = [vp, vp, sz]
    L.rj_replace_all.restype = i64
    L.rj_replace_all.argtypes = [vp, cp, sz, cp, sz, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz)]
    L.rj_free_text.argtypes = [ctypes.c_void_p]
    L.rj_scan_replace.restype = i64
    L.rj_scan_replace.argtypes = [vp, vp, u64, cp, u64, vp, u64, vp]
    L.rj_device_count.restype = ctypes.c_int
    L.rj_batch_separator.restype = ctypes.c_int
    L.rj_batch_separator.argtypes = [vp]
    L.rj_match_all_packed.restype = i64
    L.rj_match_all_packed.argtypes = [vp, vp, _u64p, ctypes.POINTER(sz), sz, u64, _u64p, ctypes.POINTER(_u64p)]
    L.rj_host_alloc.restype = vp
    L.rj_host_alloc.argtypes = [sz]
    L.rj_host_free.argtypes = [vp]
    _lib = L
    return L


def carry_decide(all_rows, world: int, rank: int, n_patterns: int, out, stream: int = 0) -> None:
    """rj_carry_decide on tensors: all_rows (world, P, 8) int64 on the device, out (4 P + 1) int64 (device or pinned)."""
    lib = load_library()
    _check(lib.rj_carry_decide(ctypes.c_void_p(all_rows.data_ptr()), world, rank, n_patterns, ctypes.c_void_p(out.data_ptr()),
                               ctypes.c_void_p(stream)))


BENCH_LIB = os.path.join(PKG, "librejit_bench.so")
_bench_lib = None


def stream_read_probe(d_text_ptr: int, n: int, launches: int = 10, stream: int = 0, default_policy: bool = False) -> float:
    """Average ms of a read-only kernel over device memory (the achievable ceiling of a scan): MEASUREMENT, from
    rejit_amd/librejit_bench.so (tools/probes/read_probe.hip) -- not part of the product library or its C ABI.
    The probe loads with the scans' own non-temporal policy (csrc/stream_load.h); default_policy=True: plain loads,
    what rounds 1-5 quoted as the ceiling."""
    global _bench_lib
    if _bench_lib is None:
        load_library()   # (torch's copy of the HIP runtime first)
        if not os.path.exists(BENCH_LIB):
            raise FileNotFoundError(f"{BENCH_LIB} is missing: run rejit_amd.build()")
        _bench_lib = ctypes.CDLL(BENCH_LIB)
        _bench_lib.rjb_stream_read_probe_policy.restype = ctypes.c_float
        _bench_lib.rjb_stream_read_probe_policy.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    ms = float(_bench_lib.rjb_stream_read_probe_policy(ctypes.c_void_p(d_text_ptr), n, launches, ctypes.c_void_p(stream), 1 if default_policy else 0))
    if ms < 0:
        raise RejitError(-3, "rjb_stream_read_probe failed")
    return ms


def device_count() -> int:
    return int(load_library().rj_device_count())


def _check(rc: int):
    if rc < 0:
        raise RejitError(int(rc), load_library().rj_last_error().decode("latin1"))
    return rc


class Program:
    """A compiled pattern (rj_program)."""

    def __init__(self, regexp):
        self._lib = load_library()
        if isinstance(regexp, str):
            regexp = regexp.encode("latin1")
        self.regexp = regexp
        h = ctypes.c_void_p()
        _check(self._lib.rj_compile(regexp, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.rj_program_free(h)
            self._h = None

    def info(self) -> dict:
        i = _Info()
        _check(self._lib.rj_program_info(self._h, ctypes.byref(i)))
        return {k: int(getattr(i, k)) for k, _ in _Info._fields_}

    # host-text entry points (the four JIT function pointers of the reference)
    def match_all(self, text: bytes) -> List[Tuple[int, int]]:
        spans = _u64p()
        n = _check(self._lib.rj_match_all(self._h, text, len(text), ctypes.byref(spans)))
        out = [(int(spans[2 * i]), int(spans[2 * i + 1])) for i in range(n)]
        if n:
            self._lib.rj_free_spans(spans)
        return out

    def match_all_batch(self, texts: List[bytes]) -> List[List[Tuple[int, int]]]:
        """MatchAll over many independent texts (files) in one device pass; offsets are relative to
        each text."""
        k = len(texts)
        arr = (ctypes.c_char_p * k)(*texts)
        sizes = (ctypes.c_size_t * k)(*[len(t) for t in texts])
        counts = (ctypes.c_uint64 * k)()
        spans = _u64p()
        total = _check(self._lib.rj_match_all_batch(self._h, arr, sizes, k, counts, ctypes.byref(spans)))
        out, at = [], 0
        for i in range(k):
            c = int(counts[i])
            out.append([(int(spans[2 * (at + j)]), int(spans[2 * (at + j) + 1])) for j in range(c)])
            at += c
        assert at == total
        if total:
            self._lib.rj_free_spans(spans)
        return out

    def match_all_batch_counts(self, texts: List[bytes]) -> List[int]:
        """rj_match_all_batch as a native caller pays for it -- the spans DO cross PCIe and are handed out -- but
        without turning them into Python tuples: the per-text counts only (bench.py's jrep extra: 10^8 line starts)."""
        k = len(texts)
        arr = (ctypes.c_char_p * k)(*texts)
        sizes = (ctypes.c_size_t * k)(*[len(t) for t in texts])
        counts = (ctypes.c_uint64 * k)()
        spans = _u64p()
        total = _check(self._lib.rj_match_all_batch(self._h, arr, sizes, k, counts, ctypes.byref(spans)))
        if total:
            self._lib.rj_free_spans(spans)
        return [int(counts[i]) for i in range(k)]

    def batch_table(self, texts: List[bytes]):
        """The caller's file table for rj_match_all_batch -- the arrays of pointers and sizes a native caller holds anyway
        (samples/jrep_gpu.cc builds them while it reads the tree) -- built once, so that a timed loop measures the library
        and not ctypes marshalling (2 500 texts: ~1 ms per call).  Keeps the texts alive."""
        k = len(texts)
        return {"k": k, "texts": texts, "arr": (ctypes.c_char_p * k)(*texts), "sizes": (ctypes.c_size_t * k)(*[len(t) for t in texts]),
                "counts": (ctypes.c_uint64 * k)()}

    def match_all_batch_table(self, table):
        """rj_match_all_batch over a batch_table(): the spans cross PCIe and are handed out (and freed here); returns the
        per-text counts as a numpy array (a view of the table's own count array: copy it to keep it)."""
        import numpy as np

        spans = _u64p()
        total = _check(self._lib.rj_match_all_batch(self._h, table["arr"], table["sizes"], table["k"], table["counts"], ctypes.byref(spans)))
        if total:
            self._lib.rj_free_spans(spans)
        return np.ctypeslib.as_array(table["counts"])

    def batch_separator(self) -> int:
        """The byte that ends a text inside a packed batch, or -1 (the pattern is matched text by text)."""
        return int(self._lib.rj_batch_separator(self._h))

    def match_all_packed(self, buf_ptr: int, offsets: List[int], sizes: List[int], total_bytes: int) -> List[List[Tuple[int, int]]]:
        """rj_match_all_packed over a buffer the caller laid out (a raw pointer: rj_host_alloc'ed or ordinary memory):
        per text the spans relative to the text."""
        k = len(offsets)
        off = (ctypes.c_uint64 * max(k, 1))(*offsets)
        sz_ = (ctypes.c_size_t * max(k, 1))(*sizes)
        counts = (ctypes.c_uint64 * max(k, 1))()
        spans = _u64p()
        total = _check(self._lib.rj_match_all_packed(self._h, ctypes.c_void_p(buf_ptr), off, sz_, k, total_bytes, counts, ctypes.byref(spans)))
        out, at = [], 0
        for i in range(k):
            c = int(counts[i])
            out.append([(int(spans[2 * (at + j)]), int(spans[2 * (at + j) + 1])) for j in range(c)])
            at += c
        assert at == total, (at, total)
        if total:
            self._lib.rj_free_spans(spans)
        return out

    def count(self, text: bytes) -> int:
        """Regej::MatchAllCount: rj_match_all with a NULL span list."""
        return int(_check(self._lib.rj_match_all(self._h, text, len(text), None)))

    def host_stats(self) -> dict:
        """rj_host_stats: which kernels answered this thread's last host-text call of the pattern."""
        s = _Stats()
        _check(self._lib.rj_host_stats(self._h, ctypes.byref(s), ctypes.sizeof(s)))
        return {k: (float(getattr(s, k)) if k.endswith("_ms") else int(getattr(s, k))) for k, _ in _Stats._fields_}

    def replace_all(self, text: bytes, repl: bytes) -> Tuple[int, bytes]:
        """(number of matches, new text): MatchAll + Replace, spliced on the GPU."""
        out = ctypes.c_void_p()
        out_len = ctypes.c_size_t()
        m = _check(self._lib.rj_replace_all(self._h, text, len(text), repl, len(repl), ctypes.byref(out), ctypes.byref(out_len)))
        data = ctypes.string_at(out, out_len.value)
        self._lib.rj_free_text(out)
        return int(m), data

    def replace_all_into(self, text: bytes, repl: bytes, dst: bytearray) -> Tuple[int, int]:
        """rj_replace_all_begin + rj_replace_all_fetch: the new text into a buffer of the caller's (number of matches, new length)."""
        out_len = ctypes.c_size_t()
        m = _check(self._lib.rj_replace_all_begin(self._h, text, len(text), repl, len(repl), ctypes.byref(out_len)))
        if out_len.value > len(dst):
            raise RejitError(-4, "replace_all_into: the new text has %d bytes, the buffer %d" % (out_len.value, len(dst)))
        buf = (ctypes.c_char * len(dst)).from_buffer(dst)
        _check(self._lib.rj_replace_all_fetch(self._h, ctypes.addressof(buf), len(dst)))
        return int(m), int(out_len.value)

    def match_first(self, text: bytes) -> Optional[Tuple[int, int]]:
        b, e = ctypes.c_uint64(), ctypes.c_uint64()
        r = _check(self._lib.rj_match_first(self._h, text, len(text), ctypes.byref(b), ctypes.byref(e)))
        return (int(b.value), int(e.value)) if r else None

    def match_anywhere(self, text: bytes) -> bool:
        return bool(_check(self._lib.rj_match_anywhere(self._h, text, len(text))))

    def match_full(self, text: bytes) -> bool:
        return bool(_check(self._lib.rj_match_full(self._h, text, len(text))))


class Scan:
    """Device-resident scanning (rj_scan): text stays in HBM, results stay in HBM."""

    def set_timing(self, on: bool = True) -> None:
        """rj_scan_set_timing: without the scan kernel's start event stats()["scan_ms"] reads 0 and a call is 6-9 us shorter."""
        _check(self._lib.rj_scan_set_timing(self._h, int(on)))

    def __init__(self, program: Program):
        self._lib = load_library()
        self.program = program
        h = ctypes.c_void_p()
        _check(self._lib.rj_scan_create(program._h, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.rj_scan_destroy(h)
            self._h = None

    def run(self, d_text_ptr: int, n: int, own_begin: int = 0, own_end: Optional[int] = None, carry_cur: int = 0,
            carry_prev_end: int = 0, have_prev: bool = False, stream: int = 0) -> int:
        if own_end is None:
            own_end = n + 1
        return int(_check(self._lib.rj_scan_run(self._h, ctypes.c_void_p(d_text_ptr), n, own_begin, own_end, carry_cur,
                                                carry_prev_end, int(have_prev), ctypes.c_void_p(stream))))

    def count(self, d_text_ptr: int, n: int, stream: int = 0) -> int:
        """rj_scan_count: MatchAllCount over device text (the one-kernel count when the pattern has the shape:
        stats()["count_path"] == 1, no span list)."""
        return int(_check(self._lib.rj_scan_count(self._h, ctypes.c_void_p(d_text_ptr), n, ctypes.c_void_p(stream))))

    def count_tensor(self, t, n: Optional[int] = None, stream=None) -> int:
        import torch

        assert t.dtype == torch.uint8 and t.is_contiguous() and t.is_cuda
        st = torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream
        return self.count(t.data_ptr(), int(t.numel() if n is None else n), stream=st)

    def gather_spans(self, d_text_ptr: int, n: int, offset: int, rank: int, world: int, root: int = 0, comm: Optional[int] = None,
                     allgather=None, gatherv=None, own_begin: int = 0, own_end: Optional[int] = None, stream: int = 0):
        """rj_scan_gather_spans: this rank's shard, the carry over the cuts, every rank's pairs (global offsets) gathered on
        `root` in text order.  Returns (job-wide count, the list on root / None elsewhere)."""
        args = (self._h, ctypes.c_void_p(d_text_ptr), n, own_begin, n + 1 if own_end is None else own_end, ctypes.c_int64(offset))
        if allgather is not None:
            total = _check(self._lib.rj_scan_gather_spans_via(*args, allgather, gatherv, None, rank, world, root, ctypes.c_void_p(stream)))
        else:
            total = _check(self._lib.rj_scan_gather_spans(*args, ctypes.c_void_p(comm), rank, world, root, ctypes.c_void_p(stream)))
        cnt = ctypes.c_uint64()
        ptr = self._lib.rj_scan_gathered_spans(self._h, ctypes.byref(cnt))
        if not ptr:
            return int(total), None
        k = int(cnt.value)
        buf = (ctypes.c_uint64 * max(2 * k, 1))()
        got = _check(self._lib.rj_scan_copy_gathered_spans(self._h, buf, k))   # (through the library: no HIP binding of our own)
        if got != k:
            raise RejitError(-3, "rj_scan_copy_gathered_spans returned %d pairs, %d were gathered" % (got, k))
        v = list(buf)
        return int(total), [(v[2 * i], v[2 * i + 1]) for i in range(int(cnt.value))]

    def start(self, d_text_ptr: int, n: int, stream: int = 0) -> None:
        """Enqueue a whole-text run; finish() returns its count (several scans can be in flight)."""
        _check(self._lib.rj_scan_start(self._h, ctypes.c_void_p(d_text_ptr), n, ctypes.c_void_p(stream)))

    def finish(self) -> int:
        return int(_check(self._lib.rj_scan_finish(self._h)))

    def run_tensor(self, t, n: Optional[int] = None, stream=None, **kw) -> int:
        """t: a contiguous uint8 torch tensor on the GPU."""
        import torch

        assert t.dtype == torch.uint8 and t.is_contiguous() and t.is_cuda
        st = torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream
        return self.run(t.data_ptr(), int(t.numel() if n is None else n), stream=st, **kw)

    def replace(self, d_text_ptr: int, n: int, repl: bytes, d_out_ptr: int, out_cap: int, stream: int = 0) -> int:
        """Replace the matches of the last run(); returns the new length (text stays in HBM)."""
        return int(_check(self._lib.rj_scan_replace(self._h, ctypes.c_void_p(d_text_ptr), n, repl, len(repl),
                                                    ctypes.c_void_p(d_out_ptr), out_cap, ctypes.c_void_p(stream))))

    def spans(self) -> List[Tuple[int, int]]:
        n = int(_check(self._lib.rj_scan_copy_spans(self._h, None, 0)))
        buf = (ctypes.c_uint64 * (2 * max(n, 1)))()
        _check(self._lib.rj_scan_copy_spans(self._h, buf, n))
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]

    def spans_tensor(self, device):
        """The matches of the last run as an (k, 2) int64 torch tensor on `device` (no trip through Python
        lists: what the multi-GPU gather of many matches works on)."""
        import torch

        n = int(_check(self._lib.rj_scan_copy_spans(self._h, None, 0)))
        out = torch.empty((max(n, 1), 2), dtype=torch.int64, device=device)
        if n:
            _check(self._lib.rj_scan_copy_spans(self._h, ctypes.cast(ctypes.c_void_p(out.data_ptr()), _u64p), n))
        return out[:n]

    def device_spans_ptr(self) -> int:
        return int(self._lib.rj_scan_device_spans(self._h) or 0)

    def stats(self) -> dict:
        s = _Stats()
        _check(self._lib.rj_scan_stats(self._h, ctypes.byref(s)))
        return {k: (float(getattr(s, k)) if k.endswith("_ms") else int(getattr(s, k))) for k, _ in _Stats._fields_}

    def match_full(self, d_text_ptr: int, n: int, stream: int = 0) -> bool:
        return bool(_check(self._lib.rj_scan_match_full(self._h, ctypes.c_void_p(d_text_ptr), n, ctypes.c_void_p(stream))))


class _BorrowedScan(Scan):
    """A rj_scan owned by a rj_multi (not destroyed from Python)."""

    def __init__(self, handle, program, owner):
        self._lib = load_library()
        self.program = program
        self._h = handle
        self._owner = owner

    def __del__(self):
        self._h = None


class MultiScan:
    """Several patterns over the same device-resident text (rj_multi): one pass over the text when
    every pattern has a nibble-form window set, else one pattern after the other."""

    def __init__(self, programs: List[Program]):
        self._lib = load_library()
        self.programs = list(programs)
        arr = (ctypes.c_void_p * len(programs))(*[p._h for p in programs])
        h = ctypes.c_void_p()
        _check(self._lib.rj_multi_create(arr, len(programs), ctypes.byref(h)))
        self._h = h
        self.fused = False
        self.how = 0

    def set_mode(self, mode: int) -> None:
        """0: fuse the scans when possible (default); 1: every pattern scans the text on its own, as one launch
        when possible; 2: one kernel per pattern on two alternating streams; 3: one kernel per pattern."""
        _check(self._lib.rj_multi_set_mode(self._h, mode))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.rj_multi_destroy(h)
            self._h = None

    def run(self, d_text_ptr: int, n: int, stream: int = 0, own_begin: int = 0, own_end: Optional[int] = None) -> List[int]:
        counts = (ctypes.c_uint64 * len(self.programs))()
        r = _check(self._lib.rj_multi_run_range(self._h, ctypes.c_void_p(d_text_ptr), n, own_begin,
                                                n + 1 if own_end is None else own_end, counts, ctypes.c_void_p(stream)))
        self.how = int(r)          # 1 fused scan, 2 separate scans + batched tails, 0 one by one
        self.fused = r == 1
        return [int(c) for c in counts]

    def start(self, d_text_ptr: int, n: int, stream: int = 0, own_begin: int = 0, own_end: Optional[int] = None) -> None:
        """Enqueue a run; finish() collects its counts.  Two MultiScan objects of the same patterns used alternately
        keep the device busy while the host turns a result around."""
        _check(self._lib.rj_multi_start(self._h, ctypes.c_void_p(d_text_ptr), n, own_begin, n + 1 if own_end is None else own_end,
                                        ctypes.c_void_p(stream)))

    def set_tail_stream(self, on: bool = True) -> None:
        """start() queues only the scan kernel on the caller's stream, the tails on a stream of the object's own."""
        _check(self._lib.rj_multi_set_tail_stream(self._h, int(on)))

    def set_counts_only(self, on: bool = True) -> bool:
        """rj_multi_set_counts_only: MatchAllCount semantics -- counts (and bounds) only, no span lists; True when the
        pattern set takes the one-kernel path (plane_count.hip)."""
        return bool(_check(self._lib.rj_multi_set_counts_only(self._h, int(on))))

    def set_timing(self, on: bool = True) -> None:
        """rj_multi_set_timing: without the scan kernel's start event scan_ms() reads 0 and consecutive kernels follow closer."""
        _check(self._lib.rj_multi_set_timing(self._h, int(on)))

    def order_after(self, other: Optional["MultiScan"]) -> None:
        """This object's scan kernels wait for the scan kernel of `other`'s run in flight (two objects, two streams)."""
        self._after = other   # (keeps it alive)
        _check(self._lib.rj_multi_order_after(self._h, other._h if other is not None else None))

    def finish(self) -> List[int]:
        counts = (ctypes.c_uint64 * len(self.programs))()
        r = _check(self._lib.rj_multi_finish(self._h, counts))
        self.how = int(r)
        self.fused = r == 1
        return [int(c) for c in counts]

    def device_counts(self, d_text_ptr: int, n: int, offset: int, rank: int, world: int, comm: Optional[int] = None, allgather=None,
                      own_begin: int = 0, own_end: Optional[int] = None, stream: int = 0) -> List[int]:
        """rj_multi_device_counts: run over this rank's shard, then the carry exchange between the shards, behind one call.
        `comm` = an ncclComm_t (integer handle) of RCCL, or `allgather` = an ALLGATHER_FN(ctx, d_send, d_recv, bytes, stream)
        callable.  Returns the job-wide counts (the same on every rank)."""
        counts = (ctypes.c_uint64 * len(self.programs))()
        args = (self._h, ctypes.c_void_p(d_text_ptr), n, own_begin, n + 1 if own_end is None else own_end, ctypes.c_int64(offset))
        if allgather is not None:
            fn = allgather if isinstance(allgather, ALLGATHER_FN) else ALLGATHER_FN(allgather)
            r = _check(self._lib.rj_multi_device_counts_via(*args, fn, None, rank, world, counts, ctypes.c_void_p(stream)))
        else:
            r = _check(self._lib.rj_multi_device_counts(*args, ctypes.c_void_p(comm), rank, world, counts, ctypes.c_void_p(stream)))
        self.how = int(r)
        self.fused = r == 1
        return [int(c) for c in counts]

    def scan(self, i: int) -> Scan:
        return _BorrowedScan(ctypes.c_void_p(self._lib.rj_multi_scan(self._h, i)), self.programs[i], self)

    def scan_ms(self) -> float:
        return float(self._lib.rj_multi_scan_ms(self._h))

    def bounds_rows(self, rows_tensor, offset: int = 0, first_round: bool = True, stream: int = 0) -> None:
        """rj_multi_bounds_device: this rank's rows of the carry exchange (8 integers per pattern, include/rejit_hip.h)
        into the (P, 8) int64 device tensor `rows_tensor`; queued on `stream`, nothing waits."""
        assert rows_tensor.dtype.itemsize == 8 and rows_tensor.numel() >= 8 * len(self.programs) and rows_tensor.is_contiguous()
        _check(self._lib.rj_multi_bounds_device(self._h, ctypes.c_int64(offset), int(first_round), ctypes.c_void_p(rows_tensor.data_ptr()),
                                                ctypes.c_void_p(stream)))

    def bounds(self, stream: int = 0) -> List[Optional[Tuple[int, int, int, int]]]:
        """Per pattern (first begin, first end, last begin, last end) of the last run, None without a
        match: what neighbouring shards exchange to carry the selection over a cut (rj_multi_bounds)."""
        k = len(self.programs)
        buf = (ctypes.c_uint64 * (4 * k))()
        _check(self._lib.rj_multi_bounds(self._h, buf, ctypes.c_void_p(stream)))
        none = (1 << 64) - 1
        return [None if buf[4 * i] == none else tuple(int(buf[4 * i + j]) for j in range(4)) for i in range(k)]
