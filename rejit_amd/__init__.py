"""rejit_amd -- MI355X-native regex scan engine behind the rejit.h API.

The product is the C-ABI shared library rejit_amd/librejit_hip.so (include/rejit_hip.h,
include/rejit.h); this package only builds it (hipcc, gfx950) and binds it with ctypes for
tests and bench.py.  There is no Python or CPU implementation of the matching path: if the
library is missing, or no HIP device is present, every call fails loudly.
"""
from .api import (RejitError, Program, Scan, MultiScan, build, library_path, load_library, device_count, stream_read_probe)  # noqa: F401

# VALU operations the fused nine-pattern scan kernel (scan_windows_fused, shared-prefilter form) spends per
# text byte on the regexdna text: MEASURED (rocprofv3 --pmc SQ_INSTS_VALU, profiles/r02_pmc_sq.txt:
# 120.86 M wave-instructions per 500 MB launch = 247.5 per 1-KiB chunk and wave = 15.5 per byte and lane).
# Of these 9.1 are the streaming loop (26 to pack a lane's 16 positions + 16 x (2 v_xad + 2 v_and + 2 v_bcnt
# + 1.5 v_min), from the ISA); the rest are the exact per-pattern tests on the chunks that pass the prefilter.
FUSED_VALU_OPS_PER_BYTE = 15.5
# plane_scan<2> (the one-pass bit-plane scan, round 3) on the same text: SQ_INSTS_VALU 43.44 M wave-instructions per
# 500 MB launch x 64 lanes / 5e8 bytes (profiles/r03_pmc_sq_counters.txt) -- 4.2 in the streaming loop (135 per
# 2-KiB pair and wave), the rest appends the candidates (about four pairs in five hold one on DNA)
PLANE_VALU_OPS_PER_BYTE = 5.56
# plane_count<2> (round 5: 32 contiguous bytes per lane, candidates in an LDS ring, classified by table lookup 64 at a time):
# SQ_INSTS_VALU 28.27 M wave-instructions per 500 MB launch x 64 lanes / 5e8 bytes (profiles/r05_pmc_sq_counters.txt) with the four code
# planes read through the VGPR index mode (plane_count.hip: plane_test); 34.13 M = 4.37 with two bit planes compared against the
# bases' masks, 33.46 M = 4.28 before the recurrence was pinned half way for 8 waves per SIMD
PLANE_COUNT_VALU_OPS_PER_BYTE = 3.62
# scan_dense_walk<1,false,4> on `[a-f]+[0-9]` over random ASCII: SQ_INSTS_VALU 1.835e9 wave instructions per 5 GB launch
# x 64 lanes / 5e9 bytes (profiles/r03_pmc_sq_counters.txt); 28.6 before round 3's instruction diet, 59 before the
# lane-packed pre-steps
# round 4: dense_streams<2,2> (bit streams, dense_streams.hip) on the same pattern and text: SQ_INSTS_VALU 144.6 M wave
# instructions per 1 GB launch x 64 lanes / 1e9 bytes (gpurun_out/r04_stream_pmc2.txt; profiles/r04_pmc_sq_counters.txt
# holds the 5 GB launch)
# round 5: 651.6 M per 5 GB launch = 8.34 (profiles/r05_pmc_sq_counters.txt: the register copies at the loop's back edge are gone;
# 8.64 at the start of the round on the same counter, 9.3 was the 1 GB launch of round 4)
DENSE_VALU_OPS_PER_BYTE = 8.34
