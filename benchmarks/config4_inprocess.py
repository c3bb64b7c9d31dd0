"""BASELINE config 4 at its full size in ONE process: 40M x 768 f32 over eight shards of one store (ott_store_create_multi) —
all eight on this box's one GPU (122.9 GB of rows + 61 GB of 16-bit planes of its 288 GB), so the shards' work runs one after
the other: the figure is what ONE GPU needs for the whole of config 4, not a scaling number.  1024 queries, cosine, take(100),
per query and merged; wall per batch, the slowest shard's score phase, the exchange + merge on the first shard's GPU.

    python benchmarks/config4_inprocess.py [rows=40000000] [shards=8]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dim, nq, k = 768, 1024, 100
s = VecStore(dim, devices=[0] * shards)
s.reserve(n)
t = time.perf_counter()
s.append_random(n, 0x07735)
while not s.batch_ready() and time.perf_counter() - t < 30:
    time.sleep(0.01)
print(f"{n} x {dim} over {shards} shards: generated + planes ready in {time.perf_counter() - t:.2f} s; shards {[c for _, _, c in s.shards()]}")
Q = np.random.default_rng(1).uniform(-1, 1, (nq, dim)).astype(np.float32)
print("| mode | wall ms (median of 5) | score ms (slowest shard) | exchange us | merge us | refined | retries |")
print("|---|---|---|---|---|---|---|")
for perq in (True, False):
    w = []
    for it in range(6):
        p = s.query(Q, Metric.Cosine).take(k)
        if perq:
            p = p.per_query()
        t0 = time.perf_counter()
        p.collect_arrays()
        if it:
            w.append((time.perf_counter() - t0) * 1e3)
    st = s.last_stats
    print(f"| {'per query' if perq else 'merged'} | {np.median(w):.2f} | {st['score_ns'] / 1e6:.2f} | {st['exchange_ns'] / 1e3:.0f} | {st['merge_ns'] / 1e3:.0f} | {st['refined']} | {st['retries']} |", flush=True)
