"""First 256-query batch on a fresh 10M x 768 store: lazy plane build inside the batch (hi_prebuild = 0) against the background
build after the appends (the default).  Wall ms of the first three batches, and how long after the append the plane was ready."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = np.random.default_rng(0).uniform(-1, 1, (256, 768)).astype(np.float32)
print("| hi_prebuild | plane ready after append (ms) | batch 0 ms | batch 1 ms | batch 2 ms |")
print("|---|---|---|---|---|")
for pre in ([int(sys.argv[2])] if len(sys.argv) > 2 else (0, -1)):  # (one mode per process shows the first batch OF A PROCESS)
    s = VecStore(768)
    s.set_option("hi_prebuild", pre)
    s.reserve(rows)
    t0 = time.perf_counter()
    s.append_random(rows, 5)
    ready = None
    if pre != 0:
        while time.perf_counter() - t0 < 10:
            if s.batch_ready():
                ready = (time.perf_counter() - t0) * 1e3
                break
            time.sleep(0.001)
    t = []
    for _ in range(3):
        t1 = time.perf_counter()
        s.query(Q, Metric.Cosine).take(100).collect_arrays()
        t.append((time.perf_counter() - t1) * 1e3)
    print(f"| {pre} | {'-' if ready is None else f'{ready:.1f}'} | {t[0]:.2f} | {t[1]:.2f} | {t[2]:.2f} |", flush=True)
    s.close()
