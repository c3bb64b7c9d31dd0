"""Queries/sec from several host threads on ONE store (ott_query is re-entrant: overlapping calls run on separate
query contexts = streams).  Small corpora are latency-bound per query, so concurrency is where their throughput is."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 768
print("| rows | threads | queries/s | mean latency us |")
print("|---|---|---|---|")
for n in (10_000, 100_000, 1_000_000):
    s = VecStore(dim)
    s.append_random(n, 5)
    qs = np.random.default_rng(1).uniform(-1, 1, (64, dim)).astype(np.float32)
    for nt in (1, 2, 4, 8, 16):
        per = 400 if n <= 100_000 else 100

        def work(i):
            for j in range(per):
                s.query(qs[(i * 7 + j) % 64], Metric.Cosine).take(10).collect_arrays()

        work(0)
        ths = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        t = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t
        print(f"| {n} | {nt} | {nt * per / dt:.0f} | {dt / per * 1e6:.0f} |", flush=True)
    s.close()
