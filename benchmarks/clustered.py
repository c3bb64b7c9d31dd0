"""How the batch cascade behaves on CLUSTERED data (the synthetic benchmark corpus is i.i.d. uniform, the friendliest case for
the hi pass): rows = normalised(centre + spread * noise) around 1000 centres, queries = perturbed rows, cosine top-k.
For each cluster spread: ms per batch, how many of the queries the hi pass certified, how many went through the split pass and
how many ended on the exact path — and that the result equals the exact path's bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Metric, Path, VecStore  # noqa: E402

n, dim, nq, k = int(os.environ.get("ROWS", "1000000")), 768, 64, int(os.environ.get("K", "100"))
rng = np.random.default_rng(3)
centres = rng.normal(0, 1, (1000, dim)).astype(np.float32)
centres /= np.linalg.norm(centres, axis=1, keepdims=True)
print(f"{n} x {dim} rows in 1000 clusters, {nq} queries, cosine top-{k}")
print("| spread (noise norm / centre norm) | first batch ms | steady ms | hi pass certified | split pass certified | exact path | equals exact path |")
print("|---|---|---|---|---|---|---|")
for spread in (1.0, 0.3, 0.1, 0.03, 0.01):
    member = rng.integers(0, 1000, n)
    rows = centres[member] + (spread / np.sqrt(dim)) * rng.normal(0, 1, (n, dim)).astype(np.float32)
    rows = (rows / np.linalg.norm(rows, axis=1, keepdims=True)).astype(np.float32)
    pick = rng.integers(0, n, nq)
    Q = (rows[pick] + (0.5 * spread / np.sqrt(dim)) * rng.normal(0, 1, (nq, dim))).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    plan = lambda path: store.query(Q, Metric.Cosine).per_query().take(k).with_path(path)
    t0 = time.perf_counter()
    a, ca = plan(Path.Mfma).collect_arrays()
    first = time.perf_counter() - t0  # includes building the hi plane (and the batch image when the split pass is needed)
    st = dict(store.last_stats)       # the first batch: no back-off yet
    times = []
    for _ in range(6):
        t0 = time.perf_counter()
        plan(Path.Mfma).collect_arrays()
        times.append(time.perf_counter() - t0)
    st2 = dict(store.last_stats)
    b, cb = plan(Path.Exact).collect_arrays()
    same = ca == cb and np.array_equal(a["index"], b["index"]) and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))
    print(f"| {spread} | {first * 1e3:.1f} | {np.median(times) * 1e3:.2f} | {nq - st['refined']} | {st['refined'] - st['retries']} | {st['retries']} | {same} |  (steady state: refined {st2['refined']} retries {st2['retries']})", flush=True)
    store.close()
