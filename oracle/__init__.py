"""CPU oracle for the otters hot path — TEST INFRASTRUCTURE ONLY.

ctypes veneer over ``oracle/_build/libotters_oracle.so`` (built from otters_oracle.c by
``make -C oracle`` / ``__graft_entry__.build()``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
the product package ``otters_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

METRIC_COSINE, METRIC_EUCLIDEAN, METRIC_DOT = 0, 1, 2
TAKE_MIN, TAKE_MAX = 0, 1
CMP_NONE, CMP_LT, CMP_GT, CMP_LTE, CMP_GTE, CMP_EQ = 0, 1, 2, 3, 4, 5
OP_EQ, OP_NEQ, OP_LT, OP_LTE, OP_GT, OP_GTE = 0, 1, 2, 3, 4, 5
REDUCE_AVX, REDUCE_SEQ4 = 0, 1
TIES_LITERAL, TIES_CANONICAL = 0, 1


class Hit(C.Structure):
    _fields_ = [("index", C.c_uint64), ("score", C.c_float), ("query", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_chunks", "pruned_chunks", "evaluated_chunks", "vectors_compared")]


HIT_DTYPE = np.dtype([("index", "<u8"), ("score", "<f4"), ("query", "<u4")])
assert HIT_DTYPE.itemsize == C.sizeof(Hit) == 16


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (a checker being built is not a checker being used)."""
    out = os.path.join(_HERE, "_build", "libotters_oracle.so")
    src = os.path.join(_HERE, "otters_oracle.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


def _has_avx2() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            return " avx2" in f.read()
    except OSError:
        return False


_libs: dict = {}


def lib(fast: bool = False) -> C.CDLL:
    name = "libotters_oracle_avx2.so" if (fast and _has_avx2()) else "libotters_oracle.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(_HERE, "_build", name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    f32p, u64p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.c_void_p
    sz = C.c_size_t
    L.otto_dot.restype = C.c_float
    L.otto_dot.argtypes = [vp, vp, sz, C.c_int]
    L.otto_l2sq.restype = C.c_float
    L.otto_l2sq.argtypes = [vp, vp, sz, C.c_int]
    L.otto_cosine.restype = C.c_float
    L.otto_cosine.argtypes = [vp, vp, sz, C.c_float, C.c_float, C.c_int]
    L.otto_inv_norm.restype = C.c_float
    L.otto_inv_norm.argtypes = [vp, sz]
    L.otto_inv_norms.restype = None
    L.otto_inv_norms.argtypes = [vp, sz, sz, vp]
    L.otto_vec_query.restype = sz
    L.otto_vec_query.argtypes = [vp, vp, sz, sz, vp, sz, C.c_int, C.c_int, sz, C.c_int, C.c_float, vp, sz, C.c_int, C.c_int, vp]
    L.otto_meta_query.restype = sz
    L.otto_meta_query.argtypes = [vp, vp, sz, sz, sz, vp, sz, C.c_int, C.c_int, sz, C.c_int, C.c_float, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    for t, ct in (("i32", C.c_int32), ("i64", C.c_int64), ("f32", C.c_float), ("f64", C.c_double)):
        fn = getattr(L, f"otto_chunk_mask_{t}")
        fn.restype = None
        fn.argtypes = [vp, vp, vp, sz, C.c_int, ct, vp]
        fn = getattr(L, f"otto_rows_mask_{t}")
        fn.restype = None
        fn.argtypes = [vp, vp, sz, sz, sz, C.c_int, ct, vp]
        fn = getattr(L, f"otto_zone_stat_{t}")
        fn.restype = None
        fn.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp]
    L.otto_rand_elem.restype = C.c_float
    L.otto_rand_elem.argtypes = [C.c_uint64, C.c_uint64]
    L.otto_rand_fill.restype = None
    L.otto_rand_fill.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    L.otto_clustered_fill.restype = None
    L.otto_clustered_fill.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_float, C.c_float]
    _libs[name] = L
    return L


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack_bits(bools) -> np.ndarray:
    """bool sequence -> BitVec<usize, Lsb0> words (bit i of the sequence = bit i%64 of word i//64)."""
    b = np.asarray(bools, dtype=bool)
    n = b.size
    words = np.zeros((n + 63) // 64 if n else 1, dtype=np.uint64)
    if n:
        by = np.packbits(b, bitorder="little")
        buf = np.zeros(words.size * 8, dtype=np.uint8)
        buf[: by.size] = by
        words = buf.view("<u8").copy()
    return words


def unpack_bits(words: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:n].astype(bool)


def dot(a, b, reduce_mode=REDUCE_AVX) -> np.float32:
    a, b = _f32(a), _f32(b)
    return np.float32(lib().otto_dot(_ptr(a), _ptr(b), a.size, reduce_mode))


def l2sq(a, b, reduce_mode=REDUCE_AVX) -> np.float32:
    a, b = _f32(a), _f32(b)
    return np.float32(lib().otto_l2sq(_ptr(a), _ptr(b), a.size, reduce_mode))


def cosine(a, b, inv_a, inv_b, reduce_mode=REDUCE_AVX) -> np.float32:
    a, b = _f32(a), _f32(b)
    return np.float32(lib().otto_cosine(_ptr(a), _ptr(b), a.size, float(inv_a), float(inv_b), reduce_mode))


def inv_norms(rows) -> np.ndarray:
    rows = _f32(rows)
    if rows.ndim == 1:
        rows = rows[None, :]
    out = np.empty(rows.shape[0], dtype=np.float32)
    if rows.shape[0]:
        lib().otto_inv_norms(_ptr(rows), rows.shape[0], rows.shape[1], _ptr(out))
    return out


def vec_query(rows, queries, metric, take, k, filter_cmp=CMP_NONE, filter_thr=0.0, row_mask=None,
              reduce_mode=REDUCE_AVX, ties=TIES_LITERAL, inv=None, fast=False) -> np.ndarray:
    """VecQueryPlan::collect.  rows [n,dim], queries [nq,dim]; row_mask: bool array (may be
    shorter than n: missing bits keep the row).  Returns a HIT_DTYPE array, best first."""
    rows = _f32(rows)
    queries = _f32(queries)
    if queries.ndim == 1:
        queries = queries[None, :]
    n, dim = (rows.shape if rows.ndim == 2 else (0, queries.shape[1]))
    nq = queries.shape[0]
    if inv is None:
        inv = inv_norms(rows) if n else np.zeros(0, np.float32)
    inv = _f32(inv)
    cap = int(min(k, n * nq))
    out = np.zeros(max(cap, 1), dtype=HIT_DTYPE)
    words, bits = None, 0
    if row_mask is not None:
        rm = np.asarray(row_mask, dtype=bool)
        words, bits = pack_bits(rm), rm.size
    m = lib(fast).otto_vec_query(_ptr(rows), _ptr(inv), n, dim, _ptr(queries), nq, metric, take, int(k), filter_cmp,
                                 float(filter_thr), _ptr(words), bits, reduce_mode, ties, _ptr(out))
    return out[:m].copy()


def meta_query(rows, chunk_size, queries, metric, take, k, filter_cmp=CMP_NONE, filter_thr=0.0, chunk_mask=None,
               row_mask=None, reduce_mode=REDUCE_AVX, ties=TIES_LITERAL, n_threads=1, inv=None, fast=False):
    """MetaQueryPlan::collect score+merge block.  chunk_mask: bool[n_chunks] or None;
    row_mask: bool[n] over global rows or None.  Returns (hits, stats dict)."""
    rows = _f32(rows)
    queries = _f32(queries)
    if queries.ndim == 1:
        queries = queries[None, :]
    n, dim = (rows.shape if rows.ndim == 2 else (0, queries.shape[1]))
    nq = queries.shape[0]
    if inv is None:
        inv = inv_norms(rows) if n else np.zeros(0, np.float32)
    inv = _f32(inv)
    cap = int(min(k, n * nq))
    out = np.zeros(max(cap, 1), dtype=HIT_DTYPE)
    cm = pack_bits(chunk_mask) if chunk_mask is not None else None
    if row_mask is not None:
        rmb = np.asarray(row_mask, dtype=bool)
        assert rmb.size == n
        rm = pack_bits(rmb)
    else:
        rm = None
    st = Stats()
    m = lib(fast).otto_meta_query(_ptr(rows), _ptr(inv), n, dim, int(chunk_size), _ptr(queries), nq, metric, take, int(k),
                                  filter_cmp, float(filter_thr), _ptr(cm), _ptr(rm), reduce_mode, ties, int(n_threads),
                                  _ptr(out), C.byref(st))
    return out[:m].copy(), {f: getattr(st, f) for f, _ in Stats._fields_}


_NP = {"i32": np.int32, "i64": np.int64, "f32": np.float32, "f64": np.float64}


def chunk_mask(kind, mn, mx, non_null, op, thr, n_chunks=None) -> np.ndarray:
    mn = np.ascontiguousarray(mn, dtype=_NP[kind])
    mx = np.ascontiguousarray(mx, dtype=_NP[kind])
    nn = np.ascontiguousarray(non_null, dtype=np.uint64)
    n = mn.size if n_chunks is None else n_chunks
    out = np.zeros(max((n + 63) // 64, 1), dtype=np.uint64)
    getattr(lib(), f"otto_chunk_mask_{kind}")(_ptr(mn), _ptr(mx), _ptr(nn), n, op, thr, _ptr(out))
    return unpack_bits(out, n)


def rows_mask(kind, vals, nulls, base, length, op, thr) -> np.ndarray:
    vals = np.ascontiguousarray(vals, dtype=_NP[kind])
    words, bits = (None, 0)
    if nulls is not None:
        nb = np.asarray(nulls, dtype=bool)
        words, bits = pack_bits(nb), nb.size
    out = np.zeros(max((length + 63) // 64, 1), dtype=np.uint64)
    getattr(lib(), f"otto_rows_mask_{kind}")(_ptr(vals), _ptr(words), bits, base, length, op, thr, _ptr(out))
    return unpack_bits(out, length)


def zone_stat(kind, vals, nulls, start, end):
    vals = np.ascontiguousarray(vals, dtype=_NP[kind])
    words, bits = (None, 0)
    if nulls is not None:
        nb = np.asarray(nulls, dtype=bool)
        words, bits = pack_bits(nb), nb.size
    if kind in ("i32", "i64"):
        mn, mx = C.c_int64(), C.c_int64()
    else:
        mn, mx = C.c_double(), C.c_double()
    nn = C.c_uint64()
    getattr(lib(), f"otto_zone_stat_{kind}")(_ptr(vals), _ptr(words), bits, start, end, C.byref(mn), C.byref(mx), C.byref(nn))
    return mn.value, mx.value, nn.value


def rand_rows(first_row: int, n_rows: int, dim: int, seed: int) -> np.ndarray:
    out = np.empty((n_rows, dim), dtype=np.float32)
    if n_rows:
        lib().otto_rand_fill(_ptr(out), first_row, n_rows, dim, seed)
    return out


def clustered_rows(first_row: int, n_rows: int, dim: int, seed: int, n_clusters: int, spread: float, aniso: float = 0.0) -> np.ndarray:
    """Rows [first_row, first_row + n_rows) of the clustered synthetic corpus (bit-identical to VecStore.append_clustered)."""
    out = np.empty((n_rows, dim), dtype=np.float32)
    if n_rows:
        lib().otto_clustered_fill(_ptr(out), first_row, n_rows, dim, seed, n_clusters, float(spread), float(aniso))
    return out
