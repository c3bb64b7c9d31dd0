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


@pytest.mark.parametrize("seed", range(int(os.environ.get("OTT_MIDSIZE_SEEDS", "8"))))  # OTT_MIDSIZE_SEEDS=200 for a soak
def test_midsize_batch_path_equals_exact_path(seed):
    """Stores of 100k .. 2M rows: large enough for several row rounds, i.e. for speculative gates, the sibling-workgroup mapping
    of 512 / 1024 queries and full persistent grids — sizes the oracle cannot score in a test.  The checker here is the
    exact-order GPU path, which the rest of the suite pins to the oracle bit for bit; the batch path (every tile width, merged
    and per-query, filters, row and chunk masks, both take kinds, appended rows) must return the same hits, on device-generated
    uniform rows and on clustered ones (where the cascade falls through its levels)."""
    rng = np.random.default_rng(50_000 + seed)
    dim = int(rng.choice([8, 24, 32, 48, 64, 96, 128, 256]))
    n = int(rng.choice([100_000, 300_000, 700_000, 2_000_000]))
    n = min(n, 40_000_000 // dim)  # keeps a store under 160 MB of rows
    store = VecStore(dim)
    cs = int(rng.choice([256, 1000, 4096]))
    store.set_chunk_size(cs)
    if seed % 3 == 2:  # clustered: many rows within the hi pass's bound of each other
        centres = rng.normal(0, 1, (64, dim)).astype(np.float32)
        rows = (centres[rng.integers(0, 64, n)] + rng.normal(0, 0.05, (n, dim))).astype(np.float32)
        store.add_vectors(rows)
        qsrc = lambda m: (centres[rng.integers(0, 64, m)] + rng.normal(0, 0.05, (m, dim))).astype(np.float32)
    else:
        store.append_random(n, 9000 + seed)
        qsrc = lambda m: rng.uniform(-1, 1, (m, dim)).astype(np.float32)
    n_chunks = (n + cs - 1) // cs
    for trial in range(3):
        nq = int(rng.choice([2, 9, 16, 31, 64, 100, 128, 200, 256, 300, 512, 700, 1024]))
        queries = qsrc(nq)
        metric = Metric(int(rng.integers(0, 3)))
        k = int(rng.choice([1, 10, 50, 100, 200]))
        kind_take = ["take", "take_min", "take_max"][int(rng.integers(0, 3))]
        perq = bool(rng.integers(0, 2))
        row_mask = (rng.random(n) < 0.5) if rng.random() < 0.3 else None
        chunk_mask = (rng.random(n_chunks) < 0.6) if rng.random() < 0.3 else None
        thr = None
        if rng.random() < 0.4:
            probe = store.query(queries[0], metric).take(max(n // 100, 1)).collect_arrays()[0]
            thr = (float(probe["score"][-1]), Cmp(int(rng.choice([2, 4]))))  # Gt / Gte at the top percentile's score
        def plan(path):
            p = getattr(store.query(queries, metric), kind_take)(k)
            if thr is not None:
                p = p.filter(thr[0], thr[1])
            if row_mask is not None:
                p = p.with_row_mask(row_mask)
            if perq:
                p = p.per_query()
            rq = p.resolve()
            rq.path = int(path)
            return store._run(rq, chunk_mask=chunk_mask)
        exact, exact_counts, _ = plan(Path.Exact)
        hits, counts, stats = plan(Path.Mfma)
        ctx = (seed, trial, n, dim, nq, metric.name, kind_take, k, perq, thr, row_mask is not None, chunk_mask is not None,
               stats["refined"], stats["retries"], stats["gate_failed"])
        assert stats["path_used"] == 2, ctx
        assert list(counts) == list(exact_counts), ctx
        assert np.array_equal(hits["index"], exact["index"]), ctx
        assert np.array_equal(hits["score"].view(np.uint32), exact["score"].view(np.uint32)), ctx
        assert np.array_equal(hits["query"], exact["query"]), ctx
        if trial == 0:  # grow the store between batches: the bf16 copies have to follow
            store.append_random(int(rng.integers(1, 5000)), 777 + seed)
            n = store.len()
            n_chunks = (n + cs - 1) // cs
            row_mask = None


OPTION_SPACE = {
    # (round 5: the product library's option table; every bit of force_fallback = one of the fallback code paths forced on)
    "exact_small": [-1, 0, 2], "hi_fmt": [-1, 0, 1, 2, 2], "force_fallback": list(range(128)),
    "mfma_f32": [0, 1], "no_hi_pass": [0, 1], "no_batch_image": [0, 1], "large_k_from": [0, 64, 256, 512],
    "small_sort": [-1, 0, 1], "stage_appends": [-1, 0, 1], "hi_prebuild": [-1, 0, 1],  # round 4: rank sort of small results, staged appends, background plane
}


@pytest.mark.parametrize("seed", range(int(os.environ.get("OTT_OPTION_SEEDS", "16"))))  # OTT_OPTION_SEEDS=300 for a soak
def test_fuzz_store_options_modes_and_large_k(oracle, seed):
    """Every store option that selects a kernel variant or a cascade policy, drawn at random together with the summation order
    (wide's AVX order / the two-f32x4 fallback), the result mode (merged / per query), k up to the sort path's range and the
    batch image on or off: whatever the combination, indices, ranks, owners and f32 score bits are the oracle's."""
    rng = np.random.default_rng(77000 + seed)
    n = int(rng.choice([9, 64, 300, 1000, 4097, 12000, 40000]))
    dim = int(rng.choice([4, 8, 24, 33, 64, 128, 200, 768]))
    nq = int(rng.choice([1, 2, 4, 7, 8, 9, 16, 17, 33, 64, 100]))
    if dim >= 200 or nq > 33:
        n = min(n, 4097)
    kind = ["uniform", "quantised", "scaled", "nasty"][seed % 4]
    rows = make_data(rng, n, dim, kind)
    queries = make_data(rng, nq, dim, "uniform" if kind == "nasty" else kind)
    store = VecStore(dim)
    cs = int(rng.choice([8, 64, 1000, 4096]))
    store.set_chunk_size(cs)
    opts = {name: int(rng.choice(vals)) for name, vals in OPTION_SPACE.items() if rng.random() < 0.5}
    for name, v in opts.items():
        store.set_option(name, v)
    reduce_mode = int(rng.integers(0, 2))
    store.set_reduce_order(reduce_mode)
    if rng.random() < 0.3:
        store.set_batch_image(True)
    store.add_vectors(rows)
    n_chunks = (n + cs - 1) // cs
    for trial in range(3):
        metric = Metric(int(rng.integers(0, 3)))
        k = int(rng.choice([1, 10, 64, 100, 257, 600, 1500, 5000]))
        perq = rng.random() < 0.5
        take = ["take_min", "take_max"][int(rng.integers(0, 2))]
        plan = getattr(store.query(queries, metric), take)(k)
        if perq:
            plan = plan.per_query()
        if rng.random() < 0.4:
            probe = oracle.vec_query(rows, queries[:1], int(metric), 1, max(n // 3, 1), reduce_mode=reduce_mode)
            plan = plan.filter(float(probe["score"][-1]) if len(probe) else 0.0, Cmp(int(rng.integers(1, 5))))
        row_mask = (rng.random(n) < 0.7) if rng.random() < 0.4 else None
        chunk_mask = (rng.random(n_chunks) < 0.7) if rng.random() < 0.4 else None
        if row_mask is not None:
            plan = plan.with_row_mask(row_mask)
        rq = plan.resolve()
        if perq:
            per = [oracle.meta_query(rows, cs, queries[q:q + 1], rq.metric, rq.take, rq.k, rq.filter_cmp, rq.filter_thr, chunk_mask=chunk_mask,
                                     row_mask=row_mask, reduce_mode=reduce_mode, ties=oracle.TIES_CANONICAL)[0] for q in range(nq)]
            for q, h in enumerate(per):
                h["query"] = q
            ref, ref_counts = np.concatenate(per), [len(h) for h in per]
        else:
            ref = oracle.meta_query(rows, cs, queries, rq.metric, rq.take, rq.k, rq.filter_cmp, rq.filter_thr, chunk_mask=chunk_mask,
                                    row_mask=row_mask, reduce_mode=reduce_mode, ties=oracle.TIES_CANONICAL)[0]
        for path in (Path.Exact, Path.Mfma, Path.Auto):
            if path == Path.Mfma and (min(k, n) + 28 > 512 or dim < 8):
                continue
            rq.path = int(path)
            hits, counts, _ = store._run(rq, chunk_mask=chunk_mask)
            ctx = (seed, trial, n, dim, nq, kind, opts, reduce_mode, metric.name, take, k, perq, path.name, rq.filter_cmp, rq.filter_thr, cs)
            assert hits.shape == ref.shape, ctx
            assert np.array_equal(hits["index"], ref["index"]), ctx
            assert np.array_equal(hits["score"].view(np.uint32), ref["score"].view(np.uint32)), ctx
            assert np.array_equal(hits["query"], ref["query"]), ctx
            if perq:
                assert [int(c) for c in counts] == ref_counts, ctx
    store.close()
