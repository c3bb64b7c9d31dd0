"""Does AUTO pick the faster path?  Single query and small batches on stores of 300k .. 10M rows: wall ms by path.
    python benchmarks/auto_choice.py [dim] [rows,rows,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Metric, Path, VecStore
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 768
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [300_000, 1_000_000, 3_000_000, 10_000_000]
print(f"| rows (dim {dim}) | nq | k | exact ms | cascade ms | auto ms | auto took |")
print("|---|---|---|---|---|---|---|")
for n in sizes:
    s = VecStore(dim); s.reserve(n); s.append_random(n, 5)
    t0 = time.perf_counter()
    while not s.batch_ready() and time.perf_counter() - t0 < 5: time.sleep(0.01)
    rng = np.random.default_rng(1)
    for nq, k in ((1, 10), (1, 100), (2, 10), (4, 10), (8, 10)):
        q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
        res = []
        for path in (Path.Exact, Path.Mfma, Path.Auto):
            ts = []
            for it in range(14):
                t = time.perf_counter(); s.query(q, Metric.Cosine).take(k).with_path(path).collect_arrays(); ts.append(time.perf_counter() - t)
            res.append(np.median(ts[3:]) * 1e3)
        took = "cascade" if s.last_stats["path_used"] == 2 else "exact"
        print(f"| {n} | {nq} | {k} | {res[0]:.3f} | {res[1]:.3f} | {res[2]:.3f} | {took} |", flush=True)
    s.close()
