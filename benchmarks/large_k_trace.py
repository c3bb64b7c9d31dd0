"""Kernel timeline of one large-k query (default take(1000), 10M x 768): run under rocprofv3 --kernel-trace by
benchmarks/large_k_trace.sh, which prints every dispatch of the last query in order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from otters_amd import Metric, VecStore
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
s = VecStore(768); s.reserve(n); s.append_random(n, 5)
q = np.random.default_rng(1).uniform(-1, 1, 768).astype(np.float32)
for it in range(3):
    t = time.perf_counter(); r = s.query(q, Metric.Cosine).take(k).collect_arrays(); dt = time.perf_counter() - t
print("k", k, "wall %.3f ms" % (dt * 1e3), "score %.3f" % (s.last_stats["score_ns"] / 1e6), "merge %.3f" % (s.last_stats["merge_ns"] / 1e6))
