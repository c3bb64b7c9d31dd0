"""A store that fills most of the GPU's 288 GB: N x 768 f32 rows (default 70M = 215 GB), generated on the device.

What is checked (the oracle cannot score 70M rows; it re-derives what is returned):
 * byte offsets far beyond 2^37: near-duplicates of the query planted at rows spread over the WHOLE store, the last row
   included, come back as the top hits, in the oracle's order, with the oracle's score bits;
 * every other returned score is the oracle's on the regenerated row (counter-based generator keyed by global row);
 * with no room for the 16-bit hi plane (107 GB beside 215 GB of rows) the batch path starts at the split-bf16 pass on the
   f32 rows: an 8-query and a 64-query batch through it equal the exact-order path, bit for bit;
 * times of the single-query pass and of the batches.

    python benchmarks/big_store.py [rows] [dim]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from otters_amd import Metric, Path, VecStore

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
SEED = 0x07735

t0 = time.time()
store = VecStore(dim)
store.reserve(n)
store.append_random(n, SEED)
store.rows(n - 1, 1)  # (waits for the generator)
print(f"{n} x {dim} f32 rows = {n * dim * 4 / 1e9:.1f} GB resident, generated in {time.time() - t0:.1f} s", flush=True)

rng = np.random.default_rng(1)
queries = oracle.rand_rows(0, 64, dim, SEED + 1)
q = queries[0]
planted = np.unique(np.concatenate([np.linspace(0, n - 1, 24).astype(np.int64), [n - 1, n - 2, min((1 << 25) + 1, n - 3), n // 2]]))
dup = (q + rng.normal(0, 0.05, (planted.size, dim))).astype(np.float32)
for i, r in zip(planted, dup):
    store.write_rows(int(i), r[None, :])


def oracle_score(row_idx, qv, metric):
    pos = np.flatnonzero(planted == row_idx)
    row = dup[pos[0]] if pos.size else oracle.rand_rows(int(row_idx), 1, dim, SEED)[0]
    if metric == Metric.Cosine:
        return oracle.cosine(qv, row, oracle.inv_norms(qv)[0], oracle.inv_norms(row)[0])
    if metric == Metric.DotProduct:
        return oracle.dot(qv, row)
    return oracle.l2sq(qv, row)


ok = True
for metric in (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
    k = planted.size + 6
    store.query(q, metric).take(k).with_path(Path.Exact).collect()
    t0 = time.perf_counter()
    res = store.query(q, metric).take(k).with_path(Path.Exact).collect()
    dt = time.perf_counter() - t0
    idx = [r.index for r in res]
    sc = np.array([r.score for r in res], np.float32)
    want = np.array([oracle_score(i, q, metric) for i in idx], np.float32)
    bits = np.array_equal(sc.view(np.uint32), want.view(np.uint32))
    # the planted rows are the best (cosine / squared L2: near-duplicates; dot: they carry the query's own norm)
    top = set(idx[:planted.size])
    planted_ok = top == set(planted.tolist())
    order_ok = bool(np.all(np.diff(sc) <= 0)) if metric != Metric.Euclidean else bool(np.all(np.diff(sc) >= 0))
    gbs = n * (dim * 4 + (4 if metric == Metric.Cosine else 0)) / dt / 1e9
    print(f"{metric.name:10s} take({k}): {dt * 1e3:7.2f} ms = {gbs:6.0f} GB/s; score bits == oracle: {bits}; planted rows (incl. row {n - 1}) are the top {planted.size}: {planted_ok}; sorted: {order_ok}", flush=True)
    ok &= bits and planted_ok and order_ok

for nq, k in ((8, 10), (64, 100)):
    for metric in (Metric.Cosine, Metric.Euclidean):
        t0 = time.perf_counter()
        a, ca = store.query(queries[:nq], metric).take(k).per_query().with_path(Path.Mfma).collect_arrays()
        dt_a = time.perf_counter() - t0
        st = dict(store.last_stats)
        t0 = time.perf_counter()
        b, cb = store.query(queries[:nq], metric).take(k).per_query().with_path(Path.Exact).collect_arrays()
        dt_b = time.perf_counter() - t0
        same = (np.array_equal(a["index"], b["index"]) and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))
                and np.array_equal(a["query"], b["query"]) and list(ca) == list(cb))
        print(f"{nq:3d} queries {metric.name:10s} top-{k} per query: batch path {dt_a * 1e3:8.2f} ms (path_used {st['path_used']}, passes {st['passes']}, refined {st['refined']}, "
              f"retries {st['retries']}) == exact-order path {dt_b * 1e3:8.2f} ms: {same}", flush=True)
        ok &= same
print("BIG STORE", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
