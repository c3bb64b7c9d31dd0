"""The reference's no-take call (src/vec.rs:213: every row, sorted) and other k > 512 results on SMALL stores: wall per call
through the Python mirror, best of 20, rank-sort path (small_sort = 1) against the radix-sort path (0).

    python benchmarks/default_take_small.py [rows] [dim]      (rocprofv3 --kernel-trace --stats -- python3 benchmarks/default_take_small.py for the kernels)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
s = VecStore(dim)
s.append_random(rows, 5)
rng = np.random.default_rng(0)
print(f"rows {rows} x dim {dim}, cosine; ms per call, best of 20")
print("| nq | k | mode | rank sort | radix sort | score kernel ms | merge ms |")
print("|---|---|---|---|---|---|---|")
for nq in (1, 4, 16):
    Q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    for k in (1000, rows):
        if k == rows and nq * rows > 16384 * 4:
            continue
        for perq in (False, True):
            res = []
            for small in (1, 0):
                s.set_option("small_sort", small)
                best = 1e9
                for _ in range(22):
                    p = s.query(Q if nq > 1 else Q[0], Metric.Cosine).take(k)
                    if perq:
                        p = p.per_query()
                    t = time.perf_counter()
                    p.collect_arrays()
                    best = min(best, time.perf_counter() - t)
                res.append(best * 1e3)
                if small == 1:
                    st = dict(s.last_stats)
            print(f"| {nq} | {k} | {'perq' if perq else 'merged'} | {res[0]:.3f} | {res[1]:.3f} | {st['score_ns'] / 1e6:.3f} | {st['merge_ns'] / 1e6:.3f} |", flush=True)
