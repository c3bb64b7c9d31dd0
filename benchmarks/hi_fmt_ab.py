"""A/B of the hi plane's element format on ONE store of ONE box: bf16 (hi_fmt = 0) against IEEE half (hi_fmt = 1), uniform
10M x 768, cosine.  The plane is rebuilt in the other format between the measurements (a few times per shape, alternating), so
clock / thermal drift and memory placement hit both alike."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Metric, VecStore  # noqa: E402

n, dim = 10_000_000, 768
s = VecStore(dim)
s.reserve(n)
s.append_random(n, 5)


def use(fmt):
    s.set_option("hi_fmt", fmt)
    s.set_batch_image(False)
    s.set_batch_image(True)
    s.prepare_batch()


print("| nq | k | bf16 score ms | half score ms | bf16 wall ms | half wall ms |")
print("|---|---|---|---|---|---|")
for nq, k in ((8, 10), (32, 10), (64, 10), (96, 10), (128, 10), (256, 10), (256, 100)):
    q = np.random.default_rng(nq).uniform(-1, 1, (nq, dim)).astype(np.float32)
    sc = {0: [], 1: []}
    wl = {0: [], 1: []}
    for rnd in range(3):
        for fmt in (0, 1):
            use(fmt)
            for it in range(4):
                t = time.perf_counter()
                s.query(q, Metric.Cosine).take(k).collect_arrays()
                dt = time.perf_counter() - t
                if it:
                    sc[fmt].append(s.last_stats["score_ns"] / 1e6)
                    wl[fmt].append(dt * 1e3)
    print(f"| {nq} | {k} | {np.median(sc[0]):.2f} | {np.median(sc[1]):.2f} | {np.median(wl[0]):.2f} | {np.median(wl[1]):.2f} |", flush=True)
