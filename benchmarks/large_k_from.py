"""Where should host-output queries switch from the register-resident lists (k <= 64 E per lane, E = 2 / 4 / 8) to the
sort path (score dump + radix sort, two phases)?  Wall time per query through the Python layer (median of 9), single query,
cosine, for stores of several sizes and k around the boundaries, with the switch forced (store option large_k_from)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Metric, VecStore

dim = 768
rng = np.random.default_rng(3)
print("| rows | k | lists (large_k_from = 512) ms | sort path (large_k_from = 64) ms | automatic ms |")
print("|---|---|---|---|---|")
for n in (10_000, 100_000, 1_000_000, 4_000_000, 10_000_000):
    s = VecStore(dim); s.reserve(n); s.append_random(n, 5)
    q = rng.uniform(-1, 1, dim).astype(np.float32)
    for k in (100, 128, 200, 256, 300, 512):
        row = []
        for frm in (512, 64, 0):
            s.set_option("large_k_from", frm)
            ts = []
            for it in range(11):
                t = time.perf_counter(); s.query(q, Metric.Cosine).take(k).collect_arrays(); ts.append(time.perf_counter() - t)
            row.append(sorted(ts[2:])[len(ts[2:]) // 2] * 1e3)
        print(f"| {n} | {k} | {row[0]:.3f} | {row[1]:.3f} | {row[2]:.3f} |", flush=True)
    s.close()
