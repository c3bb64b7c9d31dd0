"""Batch-size sweep of the AUTO path: 10M x 768 cosine top-k, nq = 1 ... 256 (wall, score-phase time, path, passes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
nqs = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 3, 4, 5, 6, 8, 9, 12, 16, 24, 32, 48, 64, 96, 128, 256]
s = VecStore(dim)
s.reserve(n)
s.append_random(n, 5)
print("| nq | path | passes | wall ms | score ms | ms / query | GB/s (alg.) | TFLOP/s |")
print("|---|---|---|---|---|---|---|---|")
for nq in nqs:
    q = np.random.default_rng(nq).uniform(-1, 1, (nq, dim)).astype(np.float32)
    best = None
    for it in range(4):
        t = time.perf_counter()
        s.query(q, Metric.Cosine).take(k).collect()
        dt = time.perf_counter() - t
        st = s.last_stats
        if it and (best is None or dt < best[0]):
            best = (dt, st["score_ns"] / 1e6, st["path_used"], st["passes"])
    dt, sc, path, passes = best
    print(f"| {nq} | {'mfma' if path == 2 else 'exact'} | {passes} | {dt * 1e3:.2f} | {sc:.2f} | {dt * 1e3 / nq:.3f} | "
          f"{n * (dim * 4 + 4) / sc / 1e6:.0f} | {2.0 * n * dim * nq / sc / 1e9:.1f} |", flush=True)
