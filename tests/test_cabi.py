"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/rejit_hip.h declares, reports parser errors without a GPU, and FAILS LOUDLY (no CPU
fallback) when asked to match without a HIP device."""
import os
import re

import pytest

import rejit_amd
from rejit_amd.api import C_ABI_SYMBOLS, RejitError
import vectors as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    rejit_amd.build()
    return rejit_amd.load_library()


def test_exports_match_header(lib):
    with open(os.path.join(ROOT, "include", "rejit_hip.h")) as fh:
        header = fh.read()
    declared = set(re.findall(r"\b(rj_[a-z_]+)\s*\(", header))
    assert declared == set(C_ABI_SYMBOLS), declared ^ set(C_ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_cpp_api_symbols_exported():
    import subprocess
    out = subprocess.check_output(["nm", "-DC", "--defined-only", rejit_amd.library_path()], text=True)
    for needle in ["rejit::Regej::Regej(char const*)", "rejit::Regej::MatchAll(char const*, unsigned long, std::vector",
                   "rejit::Regej::MatchFirst(", "rejit::Regej::MatchFull(", "rejit::Regej::MatchAnywhere(",
                   "rejit::Regej::MatchAllCount(", "rejit::Regej::ReplaceAll(", "rejit::Regej::Compile(",
                   "rejit::MatchAllCount(char const*, char const*, unsigned long)", "rejit::ReplaceAll(char const*",
                   "rejit::Replace(std::vector", "rejit::rejit_status_string"]:
        assert needle in out, needle


def test_parser_errors_need_no_gpu(lib):
    for e in V.semantics()["errors"]:
        with pytest.raises(RejitError) as ei:
            rejit_amd.Program(V.b(e["regex"]))
        assert ei.value.status == -1, e
    with pytest.raises(RejitError) as ei:
        rejit_amd.Program("a{2,1}")
    assert "Invalid repetition bounds: 2 > 1" in ei.value.message
    assert "Error parsing at index" in ei.value.message


def test_no_cpu_fallback(lib):
    """Without a HIP device the product must refuse to work rather than compute on the CPU."""
    if rejit_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RejitError) as ei:
        rejit_amd.Program("regexp")
    assert ei.value.status == -3


def test_product_does_not_reference_oracle():
    """Nothing under rejit_amd/ may include, link, import or execute anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rejit_amd")):
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip")):
                with open(os.path.join(dirpath, f), errors="ignore") as fh:
                    src = fh.read()
                assert "rejit_oracle" not in src and "librejit_ref" not in src and "oracle/" not in src, f
