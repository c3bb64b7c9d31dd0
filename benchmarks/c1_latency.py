"""BASELINE config 1 (1M x 128 f32, single query, Metric::Dot, take(10)) and the headline shape: wall per query through the
Python host layer and through the bare C ABI (ctypes, no stats -> no timing events), and the scoring kernel's own time."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Metric, VecStore  # noqa: E402
from otters_amd import _native as N  # noqa: E402

CONFIGS = [(1_000_000, 128, Metric.DotProduct, {}), (10_000_000, 768, Metric.Cosine, {})]
for rows, dim, metric, opts in CONFIGS:
    store = VecStore(dim)
    for name, v in opts.items():
        store.set_option(name, v)
    store.reserve(rows)
    store.append_random(rows, 7)
    qs = np.random.default_rng(3).uniform(-1, 1, (200, dim)).astype(np.float32)
    for i in range(10):
        store.query(qs[i], metric).take(10).collect()
    t0 = time.perf_counter()
    kern = []
    for i in range(200):
        store.query(qs[i], metric).take(10).collect()
        kern.append(store.last_stats["score_ns"] + store.last_stats["merge_ns"])
    wall_py = (time.perf_counter() - t0) / 200
    # bare C ABI, no stats
    d = N.QueryDesc()
    d.nq, d.metric, d.take, d.k, d.mode = 1, int(metric), 1, 10, 0
    out = np.empty(10, dtype=N.HIT_DTYPE)
    n_out = C.c_uint64(0)
    h = store._handle()
    lib = N.lib()
    t0 = time.perf_counter()
    for i in range(200):
        d.queries = qs[i].ctypes.data
        lib.ott_query(h, C.byref(d), N.ptr(out), 10, C.byref(n_out), None, None)
    wall_c = (time.perf_counter() - t0) / 200
    print(f"{rows}x{dim} {opts}: wall {wall_py * 1e6:.1f} us (python, with stats)  {wall_c * 1e6:.1f} us (C ABI, no stats)  kernels {np.mean(kern) / 1e3:.1f} us")
