"""MFMA batch path across dimensions (rows scaled so the corpus stays ~12 GB): TFLOP/s and GB/s of the score phase."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
print("| dim | rows | nq | wall ms | score ms | TFLOP/s | GB/s (alg.) |")
print("|---|---|---|---|---|---|---|")
for dim in (32, 64, 128, 256, 384, 768, 1536, 3072):
    n = int(12e9 / (dim * 4)) // 4096 * 4096
    s = VecStore(dim)
    s.reserve(n)
    s.append_random(n, 5)
    q = np.random.default_rng(dim).uniform(-1, 1, (nq, dim)).astype(np.float32)
    best = None
    for it in range(3):
        t = time.perf_counter()
        s.query(q, Metric.Cosine).take(10).collect_arrays()
        dt = time.perf_counter() - t
        sc = s.last_stats["score_ns"] / 1e6
        if it and (best is None or dt < best[0]):
            best = (dt, sc)
    print(f"| {dim} | {n} | {nq} | {best[0] * 1e3:.2f} | {best[1]:.2f} | {2.0 * n * dim * nq / best[1] / 1e9:.1f} | {n * (dim * 4 + 4) / best[1] / 1e6:.0f} |", flush=True)
    s.close()
