"""ctypes binding of libotters_hip.so (include/otters_hip.h).

The product path has no CPU fallback: if the HIP library is missing or no gfx950 GPU is
present, every compute call raises.  Nothing here imports ``oracle``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# OTT_LIB_PATH: load another build of the library (the diagnostic build with in-kernel stamps); development aid only
LIB_PATH = os.environ.get("OTT_LIB_PATH") or os.path.join(_CSRC, "libotters_hip.so")


class OttersError(Exception):
    """Error type of the host layer: str(e) is the reference's `Err(String)` payload."""


class Hit(C.Structure):
    _fields_ = [("index", C.c_uint64), ("score", C.c_float), ("query", C.c_uint32)]


HIT_DTYPE = np.dtype([("index", "<u8"), ("score", "<f4"), ("query", "<u4")])


class QueryDesc(C.Structure):
    _fields_ = [
        ("queries", C.c_void_p), ("nq", C.c_uint32), ("metric", C.c_uint32), ("take", C.c_uint32),
        ("filter_cmp", C.c_uint32), ("filter_thr", C.c_float), ("mode", C.c_uint32), ("k", C.c_uint64),
        ("chunk_mask", C.c_void_p), ("row_mask", C.c_void_p), ("row_mask_bits", C.c_uint64),
        ("use_device_row_mask", C.c_uint32), ("path", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("total_chunks", C.c_uint64), ("pruned_chunks", C.c_uint64), ("evaluated_chunks", C.c_uint64),
        ("vectors_compared", C.c_uint64), ("prune_ns", C.c_uint64), ("score_ns", C.c_uint64), ("merge_ns", C.c_uint64),
        ("total_ns", C.c_uint64), ("bytes_scanned", C.c_uint64), ("path_used", C.c_uint32), ("passes", C.c_uint32),
        ("rescored", C.c_uint64), ("retries", C.c_uint32), ("refined", C.c_uint32),
        ("err_ratio_max", C.c_float), ("gate_failed", C.c_uint32), ("bound_violations", C.c_uint32), ("i8_refined", C.c_uint32),
        ("exchange_ns", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


class Leaf(C.Structure):
    _fields_ = [("column", C.c_uint32), ("op", C.c_uint32), ("clause", C.c_uint32), ("reserved", C.c_uint32),
                ("lit_i64", C.c_int64), ("lit_f64", C.c_double)]


# ott_allgather_fn: int (*)(void* user, const void* send, void* recv, uint64_t bytes)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
COMM_ID_BYTES = 128
ABI_VERSION = 4

_lib = None


def build(force: bool = False) -> str:
    """Compile libotters_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h", ".inc"))]
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "otters_hip.h"))
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _CSRC, "-j", "8", "-s"])
    return LIB_PATH


def _preload_torch_hip() -> None:
    """PyTorch-ROCm bundles its own HIP runtime under the same soname (libamdhip64.so.7) as /opt/rocm's.  Whichever is
    loaded first serves the whole process; if it is /opt/rocm's (this library used before `import torch`), torch then
    pairs it with its own bundled HSA runtime and reports "No HIP GPUs are available".  So when torch is installed, its
    copy is loaded first, whatever the import order (torch itself is NOT imported here)."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        cand = os.path.join(libdir, "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- best effort: without torch the system runtime is the only one
        pass


def preload_torch_rccl() -> None:
    """The RCCL bundled with PyTorch-ROCm is built against torch's bundled HIP runtime (the one _preload_torch_hip makes
    the process-wide one).  ott_comm_create dlopens "librccl.so.1" by soname, so when torch is installed it is imported
    first — only when an RCCL comm is actually asked for — and the library then gets the copy torch has already loaded,
    in torch's own load order.  (Loading torch's librccl.so by hand and importing torch LATER in the same process ended in
    a double free at interpreter exit: measured on the GPU box.)  Without torch the system RCCL and HIP runtime pair up."""
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001 -- no torch: the system librccl.so.1 is found through the default search path
        pass


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OttersError(
            f"libotters_hip.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    _preload_torch_hip()
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    sig = {
        "ott_abi_version": (i32, []),
        "ott_last_error": (C.c_char_p, []),
        "ott_device_count": (i32, [vp]),
        "ott_store_create": (i32, [u32, i32, vp]),
        "ott_store_create_multi": (i32, [u32, u32, vp, vp]),
        "ott_multi_plan": (i32, [u64, u64, u32, vp]),
        "ott_store_shard_count": (i32, [vp]),
        "ott_store_shard_info": (i32, [vp, u32, vp, vp, vp]),
        "ott_store_transport": (C.c_char_p, [vp]),
        "ott_store_destroy": (i32, [vp]),
        "ott_store_reserve": (i32, [vp, u64]),
        "ott_store_append": (i32, [vp, vp, u64]),
        "ott_store_append_device": (i32, [vp, vp, u64]),
        "ott_store_append_random": (i32, [vp, u64, u64]),
        "ott_store_append_clustered": (i32, [vp, u64, u64, u32, C.c_float, C.c_float]),
        "ott_store_write_rows": (i32, [vp, u64, vp, u64]),
        "ott_store_len": (u64, [vp]),
        "ott_store_dim": (u32, [vp]),
        "ott_store_device": (i32, [vp]),
        "ott_store_set_chunk_size": (i32, [vp, u64]),
        "ott_store_set_base_offset": (i32, [vp, u64]),
        "ott_store_set_reduce_order": (i32, [vp, u32]),
        "ott_store_set_batch_image": (i32, [vp, i32]),
        "ott_store_prepare_batch": (i32, [vp]),
        "ott_store_batch_ready": (i32, [vp]),
        "ott_store_read_rows": (i32, [vp, u64, u64, vp]),
        "ott_store_read_inv_norms": (i32, [vp, u64, u64, vp]),
        "ott_store_add_column": (i32, [vp, u32, vp, vp, u64, vp]),
        "ott_store_eval_row_mask": (i32, [vp, vp, u32, u32, vp]),
        "ott_store_zone_stats": (i32, [vp, u32, u64, vp, vp, vp]),
        "ott_query": (i32, [vp, vp, vp, u64, vp, vp, vp]),
        "ott_query_device": (i32, [vp, vp, vp, u64, vp, vp]),
        "ott_store_sync": (i32, [vp]),
        "ott_store_stream": (vp, [vp]),
        "ott_merge_hits_device": (i32, [vp, vp, u64, u64, u32, u64, vp, vp]),
        "ott_merge_hits_device_grouped": (i32, [vp, vp, u64, u64, u64, u32, u64, vp, vp, vp]),
        "ott_store_set_option": (i32, [vp, C.c_char_p, C.c_int64]),
        "ott_comm_unique_id": (i32, [vp]),
        "ott_comm_create": (i32, [vp, i32, i32, i32, vp]),
        "ott_comm_create_host": (i32, [i32, i32, ALLGATHER_FN, vp, vp]),
        "ott_comm_destroy": (i32, [vp]),
        "ott_comm_rank": (i32, [vp]),
        "ott_comm_world": (i32, [vp]),
        "ott_comm_transport": (C.c_char_p, [vp]),
        "ott_comm_info": (i32, [vp, vp, vp]),
        "ott_comm_set_timeout_ms": (i32, [vp, C.c_int64]),
        "ott_comm_all_gather_host": (i32, [vp, vp, vp, u64]),
        "ott_query_sharded": (i32, [vp, vp, vp, vp, u64, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.ott_abi_version() != ABI_VERSION:
        raise OttersError("libotters_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().ott_last_error()
        err = OttersError(msg.decode("utf-8", "replace") if msg else f"libotters_hip error {rc}")
        err.status = int(rc)  # ott_status (include/otters_hip.h): -1 invalid, -2 HIP, -3 out of device memory, -4 unsupported
        raise err


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack_bits(bools) -> np.ndarray:
    """bool sequence -> BitVec<usize, Lsb0> words."""
    b = np.ascontiguousarray(bools, dtype=bool).ravel()
    n = b.size
    nwords = max((n + 63) // 64, 1)
    buf = np.zeros(nwords * 8, dtype=np.uint8)
    if n:
        by = np.packbits(b, bitorder="little")
        buf[: by.size] = by
    return buf.view("<u8")


def unpack_bits(words: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:n].astype(bool)
