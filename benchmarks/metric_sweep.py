"""The three metrics through the batch path at C2 scale (10M x 768, 256 queries, top-100): time and how many queries the
certification sent back to the exact path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

n, dim, nq = 10_000_000, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = VecStore(dim)
s.reserve(n)
s.append_random(n, 5)
Q = np.random.default_rng(1).uniform(-1, 1, (nq, dim)).astype(np.float32)
print("| metric | k | wall ms | score ms | finalize ms | queries re-run on the exact path |")
print("|---|---|---|---|---|---|")
for metric in (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
    for k in (10, 100):
        best = None
        for it in range(3):
            t = time.perf_counter()
            s.query(Q, metric).take(k).collect_arrays()
            dt = time.perf_counter() - t
            st = s.last_stats
            if it and (best is None or dt < best[0]):
                best = (dt, st["score_ns"] / 1e6, st["merge_ns"] / 1e6, st["retries"])
        print(f"| {metric.name} | {k} | {best[0] * 1e3:.2f} | {best[1]:.2f} | {best[2]:.3f} | {best[3]} |", flush=True)
