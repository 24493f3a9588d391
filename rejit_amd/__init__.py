"""rejit_amd -- MI355X-native regex scan engine behind the rejit.h API.

The product is the C-ABI shared library rejit_amd/librejit_hip.so (include/rejit_hip.h,
include/rejit.h); this package only builds it (hipcc, gfx950) and binds it with ctypes for
tests and bench.py.  There is no Python or CPU implementation of the matching path: if the
library is missing, or no HIP device is present, every call fails loudly.
"""
from .api import (RejitError, Program, Scan, MultiScan, build, library_path, load_library, device_count, stream_read_probe)  # noqa: F401

# (Round 6: the VALU lane-operations per text byte that bench.py's roofline_valu quotes are no longer constants typed in here
# -- they went stale with every kernel edit -- but read from profiles/pmc_traffic.json, where tools/collect_profiles.py puts
# what the SQ_INSTS_VALU pass of tools/profile_round.sh measured.  History of those figures: fused nine-pattern scan 15.5
# (round 2), plane_scan<2> 5.56 (round 3), dense_streams<2,2> 9.3 -> 8.34 (rounds 4-5), plane_count<2> 4.37 -> 3.62 (round 5).)
