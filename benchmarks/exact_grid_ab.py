"""Workgroups per CU of exact_kernel's persistent grid (experiment builds: otters_amd/csrc/variants/build_exact.sh, OTT_LIB_PATH): the
kernel's instantiations beyond config 1 — the headline (cosine), 2 and 4 queries per pass, k = 100 and 500 (E = 2, 8), squared L2, a
score filter, AUTO's single-query int8 sweep — on 10M x 768 and 1M x 128.  Score-kernel time (hipEvents), median of 30 calls."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Cmp, Metric, Path, VecStore  # noqa: E402

for rows, dim in ((10_000_000, 768), (1_000_000, 128)):
    store = VecStore(dim)
    store.reserve(rows)
    store.append_random(rows, 7)
    store.prepare_batch()
    rng = np.random.default_rng(1)
    shapes = [("cosine top-10, 1 query", Metric.Cosine, 1, 10, Path.Exact, None), ("cosine top-10, 2 queries", Metric.Cosine, 2, 10, Path.Exact, None),
              ("cosine top-10, 4 queries", Metric.Cosine, 4, 10, Path.Exact, None), ("cosine top-100, 1 query", Metric.Cosine, 1, 100, Path.Exact, None),
              ("cosine top-500, 1 query", Metric.Cosine, 1, 500, Path.Exact, None), ("squared L2 nearest 10, 1 query", Metric.Euclidean, 1, 10, Path.Exact, None),
              ("cosine top-10 with filter > 0.05, 1 query", Metric.Cosine, 1, 10, Path.Exact, 0.05), ("AUTO (int8 sweep), cosine top-10, 1 query", Metric.Cosine, 1, 10, Path.Auto, None)]
    for name, metric, nq, k, path, thr in shapes:
        ts, ws = [], []
        for i in range(34):
            q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
            plan = store.query(q if nq > 1 else q[0], metric)
            if thr is not None:
                plan = plan.filter(thr, Cmp.Gt)
            plan = (plan.take_min(k) if metric == Metric.Euclidean else plan.take(k)).with_path(path)
            plan.collect_arrays()
            if i >= 4:
                ts.append(store.last_stats["score_ns"] / 1e3)
                ws.append(store.last_stats["total_ns"] / 1e3)
        print(f"{rows}x{dim} {name}: score kernel {np.median(ts):.1f} us, call {np.median(ws):.1f} us (path {store.last_stats['path_used']})", flush=True)
    store.close()
