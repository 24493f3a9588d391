"""rejit_amd -- MI355X-native regex scan engine behind the rejit.h API.

The product is the C-ABI shared library rejit_amd/librejit_hip.so (include/rejit_hip.h,
include/rejit.h); this package only builds it (hipcc, gfx950) and binds it with ctypes for
tests and bench.py.  There is no Python or CPU implementation of the matching path: if the
library is missing, or no HIP device is present, every call fails loudly.
"""
from .api import (RejitError, Program, Scan, MultiScan, build, library_path, load_library, device_count)  # noqa: F401

# VALU operations the fused nine-pattern scan kernel (scan_windows_fused, shared-prefilter form) spends per
# text byte in its streaming loop, counted from its ISA: 26 to pack a lane's 16
# positions + 16 x (2 v_xad + 2 v_and + 2 v_bcnt + 1.5 v_min) = 146 per 16 bytes.  The exact per-pattern
# tests on the chunks that pass the prefilter (about every second one on DNA) come on top, so the VALU
# roofline bench.py derives from this number is a lower bound of the kernel's VALU utilisation.
FUSED_VALU_OPS_PER_BYTE = 9.1
