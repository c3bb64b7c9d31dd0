"""Rows f1 / f2 at the headline row count: the row-mask evaluator (ott_store_eval_row_mask) and the zone-statistics builder
(ott_store_zone_stats) over HBM-resident metadata columns of 10M rows, timed through the C ABI (the host wait and the D2H of
the small outputs included), beside the bytes each must read.  The vectors are dim 8: only the metadata columns matter here.

usage: python benchmarks/meta_kernels.py [rows]
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import VecStore
from otters_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
chunk = 1024
rng = np.random.default_rng(0)
store = VecStore(8)
store.add_vectors(rng.standard_normal((n, 8), dtype=np.float32))
h = store._handle()
lib = N.lib()


def add(dt, vals, nulls):
    cid = C.c_uint32(0)
    N.check(lib.ott_store_add_column(h, dt, N.ptr(vals), N.ptr(nulls), vals.size, C.byref(cid)))
    return cid.value


nulls = N.pack_bits(rng.random(n) < 0.05)
cols = {
    "i32": (add(0, rng.integers(0, 1000, n, dtype=np.int32), None), 4, 0),
    "i64+nulls": (add(1, rng.integers(0, 1 << 40, n, dtype=np.int64), nulls), 8, n // 8),
    "f32": (add(2, rng.standard_normal(n, dtype=np.float32), None), 4, 0),
    "f64+nulls": (add(3, rng.standard_normal(n), nulls), 8, n // 8),
}
n_chunks = (n + chunk - 1) // chunk


def timed(fn, reps=50):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps


print(f"{n} rows, chunk {chunk}")
print("| call | bytes read | wall per call | GB/s |")
print("|---|---|---|---|")
for name, (cid, esz, nb) in cols.items():
    is_f = name.startswith("f")
    mn = np.zeros(n_chunks, dtype=np.float64 if is_f else np.int64)
    mx = np.zeros_like(mn)
    nn = np.zeros(n_chunks, dtype=np.uint64)
    dt = timed(lambda: N.check(lib.ott_store_zone_stats(h, cid, chunk, N.ptr(mn), N.ptr(mx), N.ptr(nn))))
    b = n * esz + nb
    print(f"| zone_stats {name} | {b / 1e6:.0f} MB | {dt * 1e6:.0f} us | {b / dt / 1e9:.0f} |")


def leaves(spec):
    arr = (N.Leaf * len(spec))()
    for i, (col, op, clause, li, lf) in enumerate(spec):
        arr[i].column, arr[i].op, arr[i].clause, arr[i].lit_i64, arr[i].lit_f64 = cols[col][0], op, clause, li, lf
    return arr


# CmpOp: Eq 0, Neq 1, Lt 2, Lte 3, Gt 4, Gte 5 (include/otters_hip.h)
cases = {
    "1 leaf: i32 < 500": [("i32", 2, 0, 500, 0.0)],
    "2 clauses: i32 < 500 AND f64 > 0": [("i32", 2, 0, 500, 0.0), ("f64+nulls", 4, 1, 0, 0.0)],
    "(i32 < 100 OR f32 > 1) AND i64 >= 2^39 AND f64 <= 0.5": [("i32", 2, 0, 100, 0.0), ("f32", 4, 0, 0, 1.0), ("i64+nulls", 5, 1, 1 << 39, 0.0),
                                                             ("f64+nulls", 3, 2, 0, 0.5)],
}
for name, spec in cases.items():
    arr = leaves(spec)
    ncl = len({s[2] for s in spec})
    dt = timed(lambda: N.check(lib.ott_store_eval_row_mask(h, arr, len(spec), ncl, None)))
    b = sum(n * cols[s[0]][1] + cols[s[0]][2] for s in spec) + n // 8
    print(f"| eval_row_mask {name} | {b / 1e6:.0f} MB | {dt * 1e6:.0f} us | {b / dt / 1e9:.0f} |")
