"""Where a batch's wall time goes outside the GPU: Python wall against the C call's own total (stats.total_ns) against the score
and re-score phases (hipEvents).  Config 2 and a config-4 shard."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Metric, VecStore
print("| rows | queries | mode | wall ms | C call total ms | score ms | re-score ms | C-side host ms | Python ms |")
print("|---|---|---|---|---|---|---|---|---|")
for n, nq, k in ((10_000_000, 256, 100), (5_000_000, 1024, 100)):
    s = VecStore(768); s.reserve(n); s.append_random(n, 5)
    q = np.random.default_rng(1).uniform(-1, 1, (nq, 768)).astype(np.float32)
    for mode in ("merged", "per-query"):
        plan = s.query(q, Metric.Cosine).take(k)
        if mode == "per-query": plan = plan.per_query()
        for it in range(3): plan.collect_arrays()
        w, tot, sc, mg = [], [], [], []
        for it in range(9):
            t = time.perf_counter(); plan.collect_arrays(); w.append((time.perf_counter() - t) * 1e3)
            st = s.last_stats; tot.append(st["total_ns"] / 1e6); sc.append(st["score_ns"] / 1e6); mg.append(st["merge_ns"] / 1e6)
        W, T, S, M = np.median(w), np.median(tot), np.median(sc), np.median(mg)
        print(f"| {n} | {nq} | {mode} | {W:.3f} | {T:.3f} | {S:.3f} | {M:.3f} | {T - S - M:.3f} | {W - T:.3f} |", flush=True)
    s.close()
