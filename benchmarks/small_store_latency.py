"""Single-query latency on small stores, per exact-kernel variant (store option exact_small: 0 streaming kernel, 1 one-wave
LDS-DMA variant, 2 rows8 = eight lanes per row): wall through the bare C ABI (no stats: no timing events) and the scoring
kernel's own time, cosine top-K (default 10).

    python benchmarks/small_store_latency.py [dims] [K]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Metric, VecStore  # noqa: E402
from otters_amd import _native as N  # noqa: E402

dims = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [768]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10  # take(K)
print("| rows x dim | variant | wall us (C ABI) | score kernel us | merge us |")
print("|---|---|---|---|---|")
for dim in dims:
    for rows in (1_000, 4_000, 10_000, 16_000, 32_000, 65_000, 131_000):
        for variant in (0, 1, 2):
            if variant and (rows + 63) // 64 > 1024:
                continue
            store = VecStore(dim)
            store.set_option("exact_small", variant)
            try:
                store.append_random(rows, 7)
            except N.OttersError:  # (variant 1, round 2's one-wave kernel, exists in the diagnostic build only since round 5)
                continue
            qs = np.random.default_rng(3).uniform(-1, 1, (300, dim)).astype(np.float32)
            kern, mrg = [], []
            for i in range(60):
                store.query(qs[i], Metric.Cosine).take(K).collect()
                if i >= 10:
                    kern.append(store.last_stats["score_ns"])
                    mrg.append(store.last_stats["merge_ns"])
            d = N.QueryDesc()
            d.nq, d.metric, d.take, d.k, d.mode = 1, int(Metric.Cosine), 1, K, 0
            out = np.empty(K, dtype=N.HIT_DTYPE)
            n_out = C.c_uint64(0)
            h = store._handle()
            lib = N.lib()
            t0 = time.perf_counter()
            for i in range(300):
                d.queries = qs[i].ctypes.data
                lib.ott_query(h, C.byref(d), N.ptr(out), K, C.byref(n_out), None, None)
            wall = (time.perf_counter() - t0) / 300
            print(f"| {rows} x {dim} | {variant} | {wall * 1e6:.1f} | {np.median(kern) / 1e3:.1f} | {np.median(mrg) / 1e3:.1f} |", flush=True)
            store.close()
