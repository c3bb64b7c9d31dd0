"""Exact vs MFMA path on small corpora (where does the batch path start to pay?): wall ms per batch.

    python benchmarks/small_corpus.py [dim] [rows,rows,...] [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, Path, VecStore

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 768
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10  # take(K), merged
print("| rows | nq | exact ms | mfma ms | auto ms | auto path |")
print("|---|---|---|---|---|---|")
for n in (10_000, 20_000, 50_000, 100_000, 300_000, 1_000_000, 3_000_000) if len(sys.argv) < 3 else [int(x) for x in sys.argv[2].split(',')]:
    s = VecStore(dim)
    s.append_random(n, 5)
    for nq in (2, 4, 8, 16, 32, 64, 256):
        q = np.random.default_rng(nq).uniform(-1, 1, (nq, dim)).astype(np.float32)
        res = []
        for path in (Path.Exact, Path.Mfma, Path.Auto):
            best = 1e9
            for it in range(4):
                t = time.perf_counter()
                s.query(q, Metric.Cosine).take(K).with_path(path).collect_arrays()
                best = min(best, time.perf_counter() - t)
            res.append(best * 1e3)
        print(f"| {n} | {nq} | {res[0]:.3f} | {res[1]:.3f} | {res[2]:.3f} | {'mfma' if s.last_stats['path_used'] == 2 else 'exact'} |", flush=True)
    s.close()
