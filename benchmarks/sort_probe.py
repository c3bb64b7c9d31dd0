"""Large-k path (score dump + device radix sort) on 10M x 768: a few take(1000) queries, for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s = VecStore(768)
s.reserve(n)
s.append_random(n, 5)
q = np.random.default_rng(1).uniform(-1, 1, (nq, 768)).astype(np.float32)
for it in range(6):
    t = time.perf_counter()
    hits, _ = s.query(q, Metric.Cosine).take(1000).collect_arrays()
    dt = time.perf_counter() - t
    print(f"wall {dt * 1e3:.2f} ms score {s.last_stats['score_ns'] / 1e6:.2f} merge {s.last_stats['merge_ns'] / 1e6:.3f}", flush=True)
