"""Squared L2 through the batch cascade with the int8 plane first (default) and with the half plane alone (hi_fmt 1), 10M x 768:
nearest-12 for 1 / 32 / 256 queries; score phase, wall, queries each level left open, equality of the two results."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, Path, VecStore

n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768
s = VecStore(dim)
s.reserve(n)
s.append_random(n, 5)
Q = np.random.default_rng(1).uniform(-1, 1, (256, dim)).astype(np.float32)
print("| queries | first plane | score ms (median of 7) | wall ms | left open by int8 | by half | re-run exact | err_ratio_max | equal |")
print("|---|---|---|---|---|---|---|---|---|")
for nq in (1, 32, 256):
    ref = None
    for name, fmt in (("half", 1), ("int8", -1)):
        s.set_option("hi_fmt", fmt)
        s.set_batch_image(False)
        s.set_batch_image(True)
        s.prepare_batch()
        run = lambda: s.query(Q[:nq], Metric.Euclidean).take_min(12).with_path(Path.Mfma).collect_arrays()[0]
        run(); run()
        sc, wl = [], []
        for _ in range(7):
            t = time.perf_counter(); hits = run(); wl.append((time.perf_counter() - t) * 1e3); sc.append(s.last_stats["score_ns"] / 1e6)
        st = s.last_stats
        eq = ""
        if ref is None: ref = hits.copy()
        else: eq = str(bool(np.array_equal(ref["index"], hits["index"]) and np.array_equal(ref["score"].view(np.uint32), hits["score"].view(np.uint32))))
        print(f"| {nq} | {name} | {np.median(sc):.3f} | {np.median(wl):.3f} | {st['i8_refined']} | {st['refined']} | {st['retries']} | {st['err_ratio_max']:.3f} | {eq} |", flush=True)
