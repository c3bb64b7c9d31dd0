"""Probe of the batch path's cascade at BASELINE config 2 and smaller batches (MI355X): ms per batch, queries refined /
re-run, with the hi pass on (default) and off (OTT_NO_HI_PASS=1 in a second process)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Metric, Path, VecStore  # noqa: E402

rows = int(os.environ.get("ROWS", "10000000"))
dim = 768
store = VecStore(dim)
store.reserve(rows)
store.append_random(rows, 0x07735)
rng = np.random.default_rng(5)
print(f"rows {rows} dim {dim} OTT_NO_HI_PASS={os.environ.get('OTT_NO_HI_PASS')}")
only = os.environ.get("ONLY_NQ")
for nq, k in ((256, 100), (256, 10), (128, 100), (64, 100), (32, 10), (16, 10), (8, 10), (1024, 100)):
    if only and int(only) != nq:
        continue
    Q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    for metric in (Metric.Cosine,) if nq != 256 or k != 100 else (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
        plan = lambda: store.query(Q, metric).per_query().take(k).with_path(Path.Mfma)
        plan().collect_arrays()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            plan().collect_arrays()
        dt = (time.perf_counter() - t0) / reps
        st = store.last_stats
        print(f"nq {nq:5d} k {k:4d} {metric.name:10s} wall {dt*1e3:8.3f} ms  score {st['score_ns']/1e6:8.3f}  finalize {st['merge_ns']/1e6:7.3f}  C-ABI total {st['total_ns']/1e6:7.3f}  refined {st['refined']}  retries {st['retries']}  passes {st['passes']}", flush=True)
