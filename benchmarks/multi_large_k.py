"""k > 512 on the in-process multi-GPU store (every shard's sorted list comes to the host, G-way merge there) against one store."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Metric, VecStore
print("| rows x dim | shards | take | nq | mode | wall ms | score ms | merge ms |")
print("|---|---|---|---|---|---|---|---|")
for n, dim in ((1_000_000, 128), (10_000_000, 128)):
    for devs in (None, [0] * 4, [0] * 8):
        s = VecStore(dim, devices=devs) if devs else VecStore(dim)
        s.reserve(n); s.append_random(n, 3)
        rng = np.random.default_rng(2)
        for nq, k, perq in ((1, 1000, False), (1, 100_000, False), (1, None, False), (4, 100_000, True)):
            if k is None and n > 2_000_000: continue
            q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
            ts = []
            for it in range(6):
                p = s.query(q if nq > 1 else q[0], Metric.Cosine)
                if perq: p = p.per_query()
                if k is not None: p = p.take(k)
                t = time.perf_counter(); a, _ = p.collect_arrays(); ts.append(time.perf_counter() - t)
            st = s.last_stats
            print(f"| {n} x {dim} | {len(devs) if devs else 1} | {k if k else 'all'} | {nq} | {'per query' if perq else 'merged'} | {np.median(ts[1:]) * 1e3:.2f} | {st['score_ns'] / 1e6:.2f} | {st['merge_ns'] / 1e6:.2f} |", flush=True)
        s.close()
