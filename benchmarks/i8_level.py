"""The int8 first cascade level (option hi_fmt = 2) against the default cascade (half hi plane first) on ONE store:
score phase, wall, how many queries each level left open, equality of the results.  python benchmarks/i8_level.py [rows] [dim] [k]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from otters_amd import Metric, Path, VecStore  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
store = VecStore(dim)
store.reserve(rows)
store.append_random(rows, 0x07735)
rng = np.random.default_rng(3)
Q = rng.uniform(-1, 1, (1024, dim)).astype(np.float32)
print(f"| rows x dim | k | queries | mode | score ms (median of 7) | wall ms | i8_refined | refined | retries | err_ratio_max | equal |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for nq in (1, 8, 32, 64, 128, 256, 1024):
    ref = None
    for mode, fmt in (("half first", 1), ("int8 first", -1)):
        store.set_option("hi_fmt", fmt)
        store.set_batch_image(False)
        store.set_batch_image(True)
        store.prepare_batch()
        q = Q[:nq]
        kk = min(k, 10) if nq == 1 else k
        hits, _ = store.query(q, Metric.Cosine).take(kk).with_path(Path.Mfma).collect_arrays()  # warm (plane, code object)
        sc, wl = [], []
        for _ in range(7):
            t0 = time.perf_counter()
            hits, _ = store.query(q, Metric.Cosine).take(kk).with_path(Path.Mfma).collect_arrays()
            wl.append((time.perf_counter() - t0) * 1e3)
            sc.append(store.last_stats["score_ns"] / 1e6)
        st = store.last_stats
        same = ""
        if ref is None:
            ref = hits.copy()
        else:
            same = str(bool(np.array_equal(ref["index"], hits["index"]) and np.array_equal(ref["score"].view(np.uint32), hits["score"].view(np.uint32))
                            and np.array_equal(ref["query"], hits["query"])))
        print(f"| {rows} x {dim} | {kk} | {nq} | {mode} | {np.median(sc):.3f} | {np.median(wl):.3f} | {st['i8_refined']} | {st['refined']} | {st['retries']} | "
              f"{st['err_ratio_max']:.3f} | {same} |", flush=True)
