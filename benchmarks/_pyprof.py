import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Metric, VecStore
s = VecStore(128); s.append_random(10000, 5)
q = np.random.default_rng(1).uniform(-1, 1, 128).astype(np.float32)
for _ in range(200): s.query(q, Metric.Cosine).take(10).collect()
t = time.perf_counter()
for _ in range(3000): s.query(q, Metric.Cosine).take(10).collect()
print("us per query:", (time.perf_counter() - t) / 3000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): s.query(q, Metric.Cosine).take(10).collect()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
