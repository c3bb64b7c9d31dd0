"""Exact path at k = 10 / 64 / 100 / 128 / 256-by-lists over store shapes: score-kernel and wall time (median of 20)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Metric, VecStore
print("| rows x dim | queries | k | score kernel us | merge us | wall us |")
print("|---|---|---|---|---|---|")
for n, dim in ((1_000_000, 128), (200_000, 768), (1_000_000, 768), (10_000_000, 768)):
    s = VecStore(dim); s.reserve(n); s.append_random(n, 5); s.set_option("large_k_from", 512)
    Q = np.random.default_rng(1).uniform(-1, 1, (4, dim)).astype(np.float32)
    for nq, k in ((1, 10), (1, 64), (1, 100), (1, 128), (1, 256), (4, 100)):
        ks, ms, ws = [], [], []
        for it in range(25):
            t = time.perf_counter()
            s.query(Q[:nq] if nq > 1 else Q[0], Metric.Cosine).take(k).with_path(1).collect_arrays()
            dt = time.perf_counter() - t
            if it >= 5: ks.append(s.last_stats["score_ns"]); ms.append(s.last_stats["merge_ns"]); ws.append(dt)
        print(f"| {n} x {dim} | {nq} | {k} | {np.median(ks) / 1e3:.1f} | {np.median(ms) / 1e3:.1f} | {np.median(ws) * 1e6:.1f} |", flush=True)
    s.close()
