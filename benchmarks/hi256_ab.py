"""A/B of the 256-query hi pass in ONE process: hi256_kernel (phase-staggered) against mfma_score_kernel<4, false, 3>
(one barrier per stage), interleaved rounds on the same store and batch; results must be identical.
  python3 benchmarks/hi256_ab.py [rows] [dim] [nq] [k] [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Metric, VecStore  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 6
store = VecStore(dim)
store.reserve(rows)
store.append_random(rows, 0x07735)
Q = np.random.default_rng(1).uniform(-1, 1, (nq, dim)).astype(np.float32)
res = {}
times = {0: [], 1: []}
wall = {0: [], 1: []}
modes = {0: {"hi256": 0}, 1: {"hi256": 1, "hi256_nt": 0, "hi256_persist": 1}, 2: {"hi256": 1, "hi256_nt": 1, "hi256_persist": 1},
         3: {"hi256": 1, "hi256_nt": 0, "hi256_persist": 0}, 4: {"hi256": 1, "hi256_nt": 1, "hi256_persist": 0}}
times = {m: [] for m in modes}
wall = {m: [] for m in modes}
for r in range(rounds + 1):
    for mode, opts in modes.items():
        for name, v in opts.items():
            store.set_option(name, v)
        t0 = time.perf_counter()
        hits, _ = store.query(Q, Metric.Cosine).take(k).per_query().collect_arrays()
        dt = time.perf_counter() - t0
        st = store.last_stats
        if r:  # round 0 warms up (hi plane build, kernel attributes)
            times[mode].append(st["score_ns"] / 1e6)
            wall[mode].append(dt * 1e3)
        res[mode] = (hits, st["refined"], st["retries"])
same = all(np.array_equal(res[0][0]["index"], res[m][0]["index"]) and np.array_equal(res[0][0]["score"].view(np.uint32), res[m][0]["score"].view(np.uint32)) for m in modes)
for mode in modes:
    t = np.array(times[mode])
    print(f"{modes[mode]}: score phase ms median {np.median(t):.3f} min {t.min():.3f} max {t.max():.3f} | wall median {np.median(wall[mode]):.3f} | refined {res[mode][1]} retries {res[mode][2]}")
print("identical results:", same)
sys.exit(0 if same else 1)
