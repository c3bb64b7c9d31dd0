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
    assert _native.lib().ott_abi_version() == _native.ABI_VERSION == 4


def test_audit_build_and_stand_in_collective_are_drop_ins():
    """test infrastructure for the multi-GPU store on a one-GPU box (tests/test_gpu_multi_modes.py): the device-affinity audit
    build exports every symbol of the header (it is the same library with its HIP calls checked) plus its two own entry points;
    the stand-in collective library exports the ten nccl* names ott_comm.hip binds.  Loading them needs no GPU."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "otters_amd", "csrc"), "-j", "8", "-s", "audit"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "fake_rccl"), "-s"])
    audit = C.CDLL(os.path.join(ROOT, "otters_amd", "csrc", "libotters_hip_audit.so"))
    for n in declared_functions() + ["ott_audit_violations", "ott_audit_selftest"]:
        assert hasattr(audit, n), f"{n} missing from the audit build"
    assert audit.ott_abi_version() == 4 and audit.ott_audit_violations() == 0
    fake = C.CDLL(os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so"))
    for n in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclCommCount", "ncclGetVersion", "ncclGroupStart",
              "ncclGroupEnd", "ncclAllGather", "ncclGetErrorString", "fake_rccl_gathers"):
        assert hasattr(fake, n), n
    v = C.c_int(0)
    assert fake.ncclGetVersion(C.byref(v)) == 0 and v.value == 9900001  # says what it is to whoever prints the version


def c_layout():
    """sizes and offsets as a C11 compiler sees include/otters_hip.h: printed by tests/c/abi_layout (pure C, -pedantic
    -Werror, full of _Static_asserts; building it IS the proof that the header is plain C)"""
    import json
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "c"), "-s"])
    out = subprocess.run([os.path.join(ROOT, "tests", "c", "abi_layout")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr  # non-zero: header and library disagree on the ABI version
    return json.loads(out.stdout)


def test_struct_layouts_match_header():
    """every ctypes struct of otters_amd/_native.py against the C compiler's view of the header, field by field"""
    from otters_amd import _native as N
    lay = c_layout()
    assert lay["abi_version_header"] == lay["abi_version_library"] == N.ABI_VERSION
    pairs = {"ott_hit": N.Hit, "ott_query_desc": N.QueryDesc, "ott_stats": N.Stats, "ott_leaf": N.Leaf}
    for cname, ct in pairs.items():
        assert C.sizeof(ct) == lay["sizeof"][cname], cname
        fields = {k.split(".", 1)[1]: v for k, v in lay["offsetof"].items() if k.startswith(cname + ".")}
        assert sorted(fields) == sorted(n for n, _ in ct._fields_), (cname, sorted(fields))
        for name, off in fields.items():
            assert getattr(ct, name).offset == off, (cname, name)
    assert N.HIT_DTYPE.itemsize == lay["sizeof"]["ott_hit"]
    assert [N.HIT_DTYPE.fields[n][1] for n in ("index", "score", "query")] == [lay["offsetof"]["ott_hit." + n] for n in ("index", "score", "query")]


def test_host_comm_all_gather_roundtrip():
    """the HOST transport of ott_comm needs no GPU: a 1-rank comm echoes, a bad callback result is reported"""
    import numpy as np
    from otters_amd import OttersError
    from otters_amd.dist import Comm
    c = Comm.host(0, 1, lambda b: b)
    assert c.transport == "host" and c.world == 1
    assert np.array_equal(c.all_gather_host(np.arange(5, dtype=np.int64)), np.arange(5, dtype=np.int64)[None, :])
    assert c.all_gather_bytes(b"abc") == [b"abc"]
    bad = Comm.host(0, 1, lambda b: b + b"x")  # wrong size
    with pytest.raises(OttersError):
        bad.all_gather_host(np.zeros(3, dtype=np.uint8))


def test_options_are_validated_without_a_gpu():
    """ott_store_set_option's argument checks come before any device work"""
    from otters_amd import _native as N
    assert N.lib().ott_store_set_option(None, b"mfma_f32", 1) != 0
    assert b"NULL" in N.lib().ott_last_error()


def test_no_gpu_fails_loudly():
    """the product path has no CPU fallback: without a GPU a compute call raises"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from otters_amd import Metric, OttersError, VecStore
    store = VecStore(3)
    with pytest.raises(OttersError):
        store.add_vector([1.0, 0.0, 0.0])
