"""BASELINE config 1 (1M x 128 f32, single query, Metric::Dot, take(10)) on the exact-order kernel, asked for by name: per-call wall
through the bare C ABI (no stats -> no timing events) and through Python with stats, scoring / merge kernel times (hipEvents), over
3 x 400 calls (medians).  OTT_LIB_PATH picks an experiment build (otters_amd/csrc/variants/build_exact.sh)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Metric, Path, VecStore  # noqa: E402
from otters_amd import _native as N  # noqa: E402

rows, dim = int(os.environ.get("OTT_N", 1_000_000)), int(os.environ.get("OTT_DIM", 128))
store = VecStore(dim)
store.set_option("hi_prebuild", 0)  # (no background plane build beside the measurement)
store.reserve(rows)
store.append_random(rows, 7)
qs = np.random.default_rng(3).uniform(-1, 1, (400, dim)).astype(np.float32)
for i in range(20):
    store.query(qs[i], Metric.DotProduct).take(10).with_path(Path.Exact).collect()
d = N.QueryDesc()
d.nq, d.metric, d.take, d.k, d.mode, d.path = 1, int(Metric.DotProduct), 1, 10, 0, int(Path.Exact)
out = np.empty(10, dtype=N.HIT_DTYPE)
n_out = C.c_uint64(0)
h, lib = store._handle(), N.lib()
wc, wp, ks, km = [], [], [], []
for rep in range(3):
    per = []
    for i in range(400):
        d.queries = qs[i].ctypes.data
        t = time.perf_counter()
        lib.ott_query(h, C.byref(d), N.ptr(out), 10, C.byref(n_out), None, None)
        per.append(time.perf_counter() - t)
    wc.append(np.median(per) * 1e6)
    per = []
    for i in range(400):
        t = time.perf_counter()
        store.query(qs[i], Metric.DotProduct).take(10).with_path(Path.Exact).collect_arrays()
        per.append(time.perf_counter() - t)
        ks.append(store.last_stats["score_ns"] / 1e3)
        km.append(store.last_stats["merge_ns"] / 1e3)
    wp.append(np.median(per) * 1e6)
assert store.last_stats["path_used"] == 1
print(f"{rows}x{dim} dot top-10, exact-order kernel: C ABI no stats {min(wc):.1f} us (medians {['%.1f' % x for x in wc]}), python with stats {min(wp):.1f} us, "
      f"score kernel {np.median(ks):.1f} us, merge {np.median(km):.1f} us")
