"""A/B of the hi plane's element format on ONE box: bf16 (hi_fmt = 0) against IEEE half (hi_fmt = 1), uniform 10M x 768,
cosine.  Alternates the two stores batch by batch so that clock / thermal drift hits both alike."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Metric, VecStore  # noqa: E402

n, dim = 10_000_000, 768
stores = {}
for fmt in (0, 1):
    s = VecStore(dim)
    s.set_option("hi_fmt", fmt)
    s.reserve(n)
    s.append_random(n, 5)
    s.prepare_batch()
    stores[fmt] = s
print("| nq | k | bf16 score ms | half score ms | bf16 wall | half wall | rescored bf16 / half |")
print("|---|---|---|---|---|---|---|")
for nq, k in ((32, 10), (64, 10), (128, 10), (256, 10), (256, 100)):
    q = np.random.default_rng(nq).uniform(-1, 1, (nq, dim)).astype(np.float32)
    res = {0: [], 1: []}
    wall = {0: [], 1: []}
    resc = {}
    for it in range(7):
        for fmt in (0, 1):
            t = time.perf_counter()
            stores[fmt].query(q, Metric.Cosine).take(k).collect_arrays()
            dt = time.perf_counter() - t
            if it:
                res[fmt].append(stores[fmt].last_stats["score_ns"] / 1e6)
                wall[fmt].append(dt * 1e3)
            resc[fmt] = stores[fmt].last_stats["rescored"]
    print(f"| {nq} | {k} | {np.median(res[0]):.2f} | {np.median(res[1]):.2f} | {np.median(wall[0]):.2f} | {np.median(wall[1]):.2f} | {resc[0]} / {resc[1]} |", flush=True)
