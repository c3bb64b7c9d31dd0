"""CPU: libotters_hip.so loads without a GPU and exports every function include/otters_hip.h
declares (no compute calls here); the ctypes struct layouts match the header."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "otters_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ott_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from otters_amd import _native
    _native.build()
    lib = C.CDLL(_native.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in otters_hip.h but not exported"
    assert _native.lib().ott_abi_version() == 1


def test_struct_layouts_match_header():
    from otters_amd import _native as N
    assert C.sizeof(N.Hit) == 16 and N.HIT_DTYPE.itemsize == 16
    assert C.sizeof(N.QueryDesc) == 72      # sizeof(ott_query_desc) on LP64 (checked with gcc)
    assert C.sizeof(N.Stats) == 96
    assert C.sizeof(N.Leaf) == 32
    assert N.QueryDesc.k.offset == 32 and N.QueryDesc.chunk_mask.offset == 40 and N.QueryDesc.path.offset == 68


def test_no_gpu_fails_loudly():
    """the product path has no CPU fallback: without a GPU a compute call raises"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from otters_amd import Metric, OttersError, VecStore
    store = VecStore(3)
    with pytest.raises(OttersError):
        store.add_vector([1.0, 0.0, 0.0])
