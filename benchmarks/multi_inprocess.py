"""The in-process multi-GPU store (ott_store_create_multi) against one single-GPU store holding the SAME rows, on this box's
one GPU: what the fan-out over N shards (host threads, per-shard streams and merges, the exchange, the cross-GPU merge) costs
per query when the scoring work itself is unchanged.  Every shard sits on GPU 0, so the shards' kernels share the one GPU:
wall(multi) - wall(single) is pure overhead of the N-way path, not a speed-up figure.

    python benchmarks/multi_inprocess.py [rows] [dim]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rng = np.random.default_rng(1)


def timed(store, q, k, perq=False, reps=12):
    best, stats = [], None
    for _ in range(reps):
        p = store.query(q, Metric.Cosine).take(k)
        if perq:
            p = p.per_query()
        t = time.perf_counter()
        p.collect_arrays()
        best.append(time.perf_counter() - t)
    st = store.last_stats
    return float(np.median(best)) * 1e3, float(np.min(best)) * 1e3, st


print(f"rows {rows} x dim {dim}, cosine; ms per call: median (min) of 12")
print("| store | nq | k | wall ms | score ms (slowest shard) | exchange us | merge us |")
print("|---|---|---|---|---|---|---|")
for shards in (1, 0, 2, 4, 8):  # 0 = plain single-GPU store
    s = VecStore(dim) if shards == 0 else VecStore(dim, devices=[0] * shards)
    s.reserve(rows)
    s.append_random(rows, 5)
    name = "single" if shards == 0 else f"multi x{shards}"
    for nq, k, perq in ((1, 10, False), (1, 100, False), (4, 10, False), (256, 100, False), (1024, 100, True)):
        if nq >= 256 and rows > 2_000_000 and shards not in (0, 8):
            continue
        q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
        timed(s, q, k, perq, reps=2)
        med, mn, st = timed(s, q, k, perq, reps=12 if nq < 256 else 5)
        print(f"| {name} | {nq} | {k} | {med:.3f} ({mn:.3f}) | {st['score_ns'] / 1e6:.3f} | {st['exchange_ns'] / 1e3:.1f} | {st['merge_ns'] / 1e3:.1f} |", flush=True)
    s.close()
