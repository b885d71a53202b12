"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/splitwise_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import PKG, ROOT

LIB = os.path.join(PKG, "splitwise_amd", "libsplitwise_hip.so")
HEADER = os.path.join(ROOT, "include", "splitwise_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sw_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = declared_symbols()
    for required in ("sw_reshape_and_cache", "sw_copy_blocks", "sw_swap_blocks", "sw_convert_fp8_e5m2",
                     "sw_paged_attention_v1", "sw_paged_attention_v2", "sw_last_error"):
        assert required in names


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_python_binding_covers_the_header():
    from splitwise_amd import _C
    assert sorted(_C.exported_symbols()) == declared_symbols()
    lib = _C.load_library()
    header = open(HEADER).read()
    declared = int(re.search(r"#define SW_ABI_VERSION (\d+)", header).group(1))
    assert lib.sw_abi_version() == declared == _C.ABI_VERSION


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_argument_errors_are_reported_without_a_gpu():
    from splitwise_amd import _C
    lib = _C.load_library()
    # exactly one side must be fp8: rejected before any HIP call
    rc = lib.sw_convert_fp8_e5m2(None, None, 8, _C.SW_F16, _C.SW_F16, None)
    assert rc == -1
    assert b"fp8" in lib.sw_last_error()
