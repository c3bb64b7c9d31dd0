"""take(k) sweep, single query, 10M x 768 cosine (exact path; k > 512 leaves the fused register top-k for the sort path)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Cmp, Metric, VecStore

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
s = VecStore(dim)
s.reserve(n)
s.append_random(n, 5)
q = np.random.default_rng(1).uniform(-1, 1, dim).astype(np.float32)
print("| k | filter | wall ms | score ms | merge ms | hits |")
print("|---|---|---|---|---|---|")
for k, flt in [(1, None), (10, None), (64, None), (100, None), (128, None), (256, None), (512, None), (1000, None), (10000, None),
               (100000, None), (10, 0.05), (1000, 0.05), (10_000_000, 0.08)]:
    best = None
    for it in range(3):
        p = s.query(q, Metric.Cosine)
        if flt is not None:
            p = p.filter(flt, Cmp.Gt)
        t = time.perf_counter()
        hits, _ = p.take(k).collect_arrays()
        dt = time.perf_counter() - t
        st = s.last_stats
        if best is None or dt < best[0]:
            best = (dt, st["score_ns"] / 1e6, st["merge_ns"] / 1e6, len(hits))
    print(f"| {k} | {flt} | {best[0] * 1e3:.2f} | {best[1]:.2f} | {best[2]:.3f} | {best[3]} |", flush=True)
