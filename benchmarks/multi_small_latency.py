"""Latency of a single top-10 query on SMALL stores: one store against the in-process multi-GPU store (all shards on GPU 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Metric, VecStore
print("| rows x dim | shards | plan | us per query (median of 300) | shard rows |")
print("|---|---|---|---|---|")
for n, dim in ((1000, 128), (10_000, 768), (100_000, 768), (1_000_000, 128)):
    rows = np.random.default_rng(0).uniform(-1, 1, (n, dim)).astype(np.float32)
    q = np.random.default_rng(1).uniform(-1, 1, dim).astype(np.float32)
    for devs in (None, [0, 0], [0] * 8):
        for plan in ((False,) if devs is None else (True, False)):
            s = VecStore(dim, devices=devs) if devs else VecStore(dim)
            if plan: s.reserve(n)
            s.add_vectors(rows)
            ts = []
            for it in range(320):
                t = time.perf_counter(); s.query(q, Metric.Cosine).take(10).collect_arrays(); ts.append(time.perf_counter() - t)
            sh = [c for _, _, c in s.shards()] if devs else [n]
            print(f"| {n} x {dim} | {len(devs) if devs else 1} | {'reserve' if plan else 'none'} | {np.median(ts[20:]) * 1e6:.1f} | {sh} |", flush=True)
            s.close()
