"""ctypes bindings for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

* `Oracle`  -- oracle/librejit_oracle.so, our plain-C restatement of the reference's
               algorithm (oracle/rejit_oracle.c); built on demand with gcc.
* `Ref`     -- oracle/_ref/librejit_ref.so, the REAL reference compiled in place from
               /root/reference by oracle/Makefile (only buildable in the build
               container; the GPU box uses the prebuilt file).  Always driven with
               use_fast_forward=0 unless a test asks for the (buggy) default flags.

Nothing under rejit_amd/ imports this module.
"""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librejit_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "librejit_ref.so")

_u64p = ctypes.POINTER(ctypes.c_uint64)


def build_oracle():
    src = os.path.join(ORACLE_DIR, "rejit_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def build_ref():
    """Build oracle/_ref from /root/reference when the sources are present."""
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref", "-j8"], stdout=subprocess.DEVNULL)
    return REF_SO if os.path.exists(REF_SO) else None


def _pairs(buf, n):
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]


class Oracle:
    PARSE_ERROR = -1
    REJECTED = -2

    def __init__(self):
        self.lib = ctypes.CDLL(build_oracle())
        L = self.lib
        L.ro_match_all_re.restype = ctypes.c_long
        L.ro_match_all_re.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, _u64p, ctypes.c_size_t]
        L.ro_match_all_spec_re.restype = ctypes.c_long
        L.ro_match_all_spec_re.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, _u64p, ctypes.c_size_t]
        L.ro_match_full_re.restype = ctypes.c_int
        L.ro_match_full_re.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        L.ro_error.restype = ctypes.c_char_p
        L.ro_compile.restype = ctypes.c_int
        L.ro_compile.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        L.ro_free.argtypes = [ctypes.c_void_p]
        L.ro_match_all.restype = ctypes.c_long
        L.ro_match_all.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, _u64p, ctypes.c_size_t]

    def status(self, regex: bytes) -> int:
        h = ctypes.c_void_p()
        st = self.lib.ro_compile(regex, ctypes.byref(h))
        if st == 0:
            self.lib.ro_free(h)
        return st

    def match_all(self, regex: bytes, text: bytes, cap=None):
        """Returns a list of (begin, end) or a negative status code."""
        cap = (len(text) + 2) if cap is None else cap
        buf = (ctypes.c_uint64 * (2 * max(cap, 1)))()
        n = self.lib.ro_match_all_re(regex, text, len(text), buf, cap)
        if n < 0:
            return int(n)
        return _pairs(buf, min(n, cap))

    def match_all_spec(self, regex: bytes, text: bytes):
        """Documented left-most-longest semantics (differs from match_all only on Q8)."""
        cap = len(text) + 2
        buf = (ctypes.c_uint64 * (2 * cap))()
        n = self.lib.ro_match_all_spec_re(regex, text, len(text), buf, cap)
        if n < 0:
            return int(n)
        return _pairs(buf, min(n, cap))

    def longest_all(self, regex: bytes, text: bytes):
        """ends[s] for s in 0..n: end of the longest match beginning at s, or -1."""
        ends = (ctypes.c_int64 * (len(text) + 1))()
        self.lib.ro_longest_all_re.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                                               ctypes.POINTER(ctypes.c_int64)]
        st = self.lib.ro_longest_all_re(regex, text, len(text), ends)
        assert st == 0, st
        return list(ends)

    def count(self, regex: bytes, text: bytes) -> int:
        return int(self.lib.ro_match_all_re(regex, text, len(text), None, 0))

    def match_full(self, regex: bytes, text: bytes):
        return int(self.lib.ro_match_full_re(regex, text, len(text)))

    def match_first(self, regex: bytes, text: bytes):
        r = self.match_all(regex, text)
        if isinstance(r, int):
            return r
        return r[0] if r else None

    def error(self) -> str:
        return self.lib.ro_error().decode("latin1")


class Ref:
    def __init__(self, use_ff=0, ff_early=None, ff_reduce=1, parser_opt=1):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        self.lib = ctypes.CDLL(REF_SO)
        L = self.lib
        L.ref_match_all.restype = ctypes.c_long
        L.ref_match_all.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, _u64p, ctypes.c_size_t]
        L.ref_match_first.restype = ctypes.c_int
        L.ref_match_first.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, _u64p]
        L.ref_match_full.restype = ctypes.c_int
        L.ref_match_full.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        L.ref_match_anywhere.restype = ctypes.c_int
        L.ref_match_anywhere.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        L.ref_match_all_repeat.restype = ctypes.c_long
        L.ref_match_all_repeat.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        self.set_flags(use_ff, use_ff if ff_early is None else ff_early, ff_reduce, parser_opt)

    def set_flags(self, use_ff, ff_early, ff_reduce, parser_opt):
        self.lib.ref_set_flags(int(use_ff), int(ff_early), int(ff_reduce), int(parser_opt))

    def match_all(self, regex: bytes, text: bytes, cap=None):
        cap = (len(text) + 2) if cap is None else cap
        buf = (ctypes.c_uint64 * (2 * max(cap, 1)))()
        n = self.lib.ref_match_all(regex, text, len(text), buf, cap)
        if n < 0:
            return int(n)
        return _pairs(buf, min(n, cap))

    def match_full(self, regex: bytes, text: bytes):
        return int(self.lib.ref_match_full(regex, text, len(text)))

    def match_first(self, regex: bytes, text: bytes):
        be = (ctypes.c_uint64 * 2)()
        r = self.lib.ref_match_first(regex, text, len(text), be)
        if r < 0:
            return int(r)
        return (int(be[0]), int(be[1])) if r else None

    def match_anywhere(self, regex: bytes, text: bytes):
        return int(self.lib.ref_match_anywhere(regex, text, len(text)))


def have_ref() -> bool:
    return os.path.exists(REF_SO)
