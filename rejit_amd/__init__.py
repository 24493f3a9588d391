"""rejit_amd -- MI355X-native regex scan engine behind the rejit.h API.

The product is the C-ABI shared library rejit_amd/librejit_hip.so (include/rejit_hip.h,
include/rejit.h); this package only builds it (hipcc, gfx950) and binds it with ctypes for
tests and bench.py.  There is no Python or CPU implementation of the matching path: if the
library is missing, or no HIP device is present, every call fails loudly.
"""
from .api import (RejitError, Program, Scan, MultiScan, build, library_path, load_library, device_count)  # noqa: F401

# VALU operations the fused nine-pattern scan kernel (scan_windows_fused) spends per text byte, counted
# from its ISA; bench.py prices the kernel's VALU roofline with it
FUSED_VALU_OPS_PER_BYTE = 29.0
