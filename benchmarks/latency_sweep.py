"""Single-query latency vs corpus size (exact path): wall and score-kernel time; OTT_EXACT_DEEP=0/1 forces the prefetch depth."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 768
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print("| rows | wall us | score us | merge us | GB/s (alg., score kernel) |")
print("|---|---|---|---|---|")
for n in (1_000, 10_000, 50_000, 100_000, 200_000, 300_000, 500_000, 1_000_000, 2_000_000, 4_000_000):
    s = VecStore(dim)
    s.append_random(n, 5)
    q = np.random.default_rng(1).uniform(-1, 1, (nq, dim)).astype(np.float32)
    best = None
    for it in range(12):
        t = time.perf_counter()
        s.query(q, Metric.Cosine).take(10).collect_arrays()
        dt = time.perf_counter() - t
        st = s.last_stats
        if it >= 2 and (best is None or dt < best[0]):
            best = (dt, st["score_ns"] / 1e3, st["merge_ns"] / 1e3)
    print(f"| {n} | {best[0] * 1e6:.0f} | {best[1]:.0f} | {best[2]:.0f} | {n * (dim * 4 + 4) / best[1] / 1e3:.0f} |", flush=True)
    s.close()
