"""GPU: randomized differential test — exact path, MFMA path and AUTO against the oracle's canonical
form on random shapes, metrics, take kinds, filters, row / chunk masks and awkward data (ties,
duplicates, zero rows, NaN / inf rows).  Everything must match bit for bit."""
import os

import numpy as np
import pytest

from otters_amd import Cmp, Metric, Path, VecStore

pytestmark = pytest.mark.gpu


def make_data(rng, n, dim, kind):
    if kind == "uniform":
        rows = rng.uniform(-1, 1, (n, dim))
    elif kind == "quantised":
        rows = rng.integers(-3, 4, (n, dim)).astype(np.float64)
    elif kind == "scaled":
        rows = rng.normal(0, 1, (n, dim)) * np.exp(rng.normal(0, 2, (n, 1)))
    else:  # nasty: zero rows, duplicates, NaN / inf
        rows = rng.uniform(-1, 1, (n, dim))
        for _ in range(max(n // 50, 1)):
            rows[rng.integers(n)] = 0.0
            rows[rng.integers(n)] = rows[rng.integers(n)]
        if n > 20:
            rows[rng.integers(n), rng.integers(dim)] = np.nan
            rows[rng.integers(n), rng.integers(dim)] = np.inf
            rows[rng.integers(n), rng.integers(dim)] = -np.inf
    return rows.astype(np.float32)


@pytest.mark.parametrize("seed", range(int(os.environ.get("OTT_FUZZ_SEEDS", "24"))))  # OTT_FUZZ_SEEDS=400 for a long soak
def test_fuzz_paths_against_oracle(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 7, 63, 64, 65, 255, 257, 1000, 5000, 20000, 70000]))
    dim = int(rng.choice([1, 3, 7, 8, 9, 16, 31, 32, 33, 64, 100, 128, 200, 768, 1000, 1030]))  # 1000 / 1030: past the in-argument query limit
    nq = int(rng.choice([1, 2, 3, 5, 8, 9, 12, 16, 17, 24, 32, 33, 48, 64, 65, 130, 257, 520, 1030]))  # every MFMA tile width; 2, 3 and 5 blocks of 256
    if dim >= 768 or nq > 300:
        n = min(n, 5000)  # keeps the oracle's share of the test short
    kind = ["uniform", "quantised", "scaled", "nasty"][seed % 4]
    rows = make_data(rng, n, dim, kind)
    queries = make_data(rng, nq, dim, "uniform" if kind == "nasty" else kind)
    if kind == "quantised" and n > 10:
        rows[3] = queries[0]
    store = VecStore(dim)
    cs = int(rng.choice([1, 3, 64, 100, 1024]))
    store.set_chunk_size(cs)
    store.add_vectors(rows)
    n_chunks = (n + cs - 1) // cs
    for trial in range(4):
        metric = Metric(int(rng.integers(0, 3)))
        k = int(rng.choice([1, 2, 10, 63, 64, 65, 100, 200, 400]))
        kind_take = ["take", "take_min", "take_max"][int(rng.integers(0, 3))]
        plan = getattr(store.query(queries, metric), kind_take)(k)
        if rng.random() < 0.5:
            probe = oracle.vec_query(rows, queries[:1], int(metric), 1, max(n // 2, 1))
            thr = float(probe["score"][-1]) if len(probe) else 0.0
            plan = plan.filter(thr, Cmp(int(rng.integers(1, 6))))
        row_mask = (rng.random(n if rng.random() < 0.7 else max(n // 2, 1)) < 0.6) if rng.random() < 0.5 else None
        chunk_mask = (rng.random(n_chunks) < 0.6) if rng.random() < 0.5 else None
        if row_mask is not None:
            plan = plan.with_row_mask(row_mask)
        rq = plan.resolve()
        full_mask = None
        if row_mask is not None:
            full_mask = np.ones(n, bool)
            full_mask[: row_mask.size] = row_mask
        ref, rstats = oracle.meta_query(rows, cs, queries, rq.metric, rq.take, rq.k, rq.filter_cmp, rq.filter_thr,
                                        chunk_mask=chunk_mask, row_mask=full_mask, ties=oracle.TIES_CANONICAL)
        for path in (Path.Exact, Path.Mfma, Path.Auto):
            if path == Path.Mfma and (min(k, n) + 28 > 512 or dim < 8):
                continue
            rq.path = int(path)
            hits, _, stats = store._run(rq, chunk_mask=chunk_mask)
            ctx = (seed, trial, n, dim, nq, kind, metric.name, kind_take, k, path.name, rq.filter_cmp, rq.filter_thr,
                   None if row_mask is None else row_mask.size, None if chunk_mask is None else int(chunk_mask.sum()), cs)
            print("CTX", ctx, "GPU", hits[:5], "REF", ref[:5])
            assert hits.shape == ref.shape, ctx
            assert np.array_equal(hits["index"], ref["index"]), ctx
            assert np.array_equal(hits["score"].view(np.uint32), ref["score"].view(np.uint32)), ctx
            assert np.array_equal(hits["query"], ref["query"]), ctx
            assert stats is None or stats["vectors_compared"] == rstats["vectors_compared"], ctx
