"""Ingest into the in-process multi-GPU store against one store (all shards on the box's one GPU): one flat host buffer, 1000-row
pieces, ONE row per call (VecStore::add_vector, src/vec.rs:357-371), with and without a plan (ott_store_reserve)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore
from otters_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rows = np.random.default_rng(0).uniform(-1, 1, (n, dim)).astype(np.float32)
q = rows[123].copy()
print(f"| what ({n} x {dim}) | shards | seconds | GB/s | rows/s | first query after (ms) |")
print("|---|---|---|---|---|---|")
for devs in (None, [0] * 4, [0] * 8):
    G = len(devs) if devs else 1
    for what in ("flat reserved", "flat no plan", "1000-row pieces reserved", "one row per call reserved", "one row per call no plan"):
        s = VecStore(dim, devices=devs) if devs else VecStore(dim)
        m = n if "one row" not in what else min(n, 200_000)
        if "reserved" in what:
            s.reserve(m)
        h = s._handle()
        t = time.perf_counter()
        if what.startswith("flat"):
            s.add_vectors(rows)
        elif what.startswith("1000"):
            for i in range(0, m, 1000):
                N.check(N.lib().ott_store_append(h, C.c_void_p(rows[i:i + 1000].ctypes.data), min(1000, m - i)))
        else:
            for i in range(m):
                N.check(N.lib().ott_store_append(h, C.c_void_p(rows[i].ctypes.data), 1))
        dt = time.perf_counter() - t
        s._n = m  # (the raw calls above went past the Python mirror's own row count)
        t = time.perf_counter()
        a, _ = s.query(q, Metric.Cosine).take(3).collect_arrays()
        dq = time.perf_counter() - t
        assert int(a["index"][0]) == 123
        print(f"| {what} | {G} | {dt:.3f} | {m * dim * 4 / 1e9 / dt:.2f} | {m / dt:.0f} | {dq * 1e3:.1f} |", flush=True)
        s.close()
