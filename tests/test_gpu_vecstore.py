"""GPU parity tests: the HIP path (through the C ABI, via the host layer) against the CPU
oracle on the same inputs.  Bar: indices, ranks AND f32 score bits identical to the oracle's
canonical form (the reference's order of operations; ties broken by row then query).
The literal collector restatement is compared modulo the tie freedom the reference leaves."""
import numpy as np
import pytest

from helpers import check_expect, load, oracle_collect, plan_from_case, same_modulo_ties
from otters_amd import Cmp, Metric, OttersError, Path, VecStore

pytestmark = pytest.mark.gpu

VEC_CASES = load("vec_store_cases.json")


def gpu_hits(plan):
    rq = plan.resolve()
    hits, counts, stats = plan.vector_store._run(rq)
    return rq, hits, counts, stats


def assert_bit_exact(hits, ref):
    assert hits.shape == ref.shape, (hits.shape, ref.shape)
    assert np.array_equal(hits["index"], ref["index"]), (hits[:10], ref[:10])
    assert np.array_equal(hits["score"].view(np.uint32), ref["score"].view(np.uint32)), (hits[:10], ref[:10])
    assert np.array_equal(hits["query"], ref["query"])


@pytest.mark.parametrize("case", [c for c in VEC_CASES if "metric" in c], ids=lambda c: c["name"])
def test_golden_cases_on_gpu(oracle, case):
    store = VecStore(case["dim"])
    exp = case["expect"]
    if case["vectors"]:
        store.add_vectors(case["vectors"])
    plan = plan_from_case(case, store)
    if "error_contains" in exp or "error_eq" in exp:
        with pytest.raises(OttersError) as ei:
            plan.collect()
        assert exp.get("error_eq", exp.get("error_contains")) in str(ei.value)
        return
    res = plan.collect()
    check_expect([r.index for r in res], [r.score for r in res], exp)
    rq = plan.resolve()
    rows = np.asarray(case["vectors"], np.float32).reshape(-1, case["dim"])
    ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
    assert [r.index for r in res] == [int(i) for i in ref["index"]]
    assert np.array_equal(np.array([r.score for r in res], np.float32).view(np.uint32), ref["score"].view(np.uint32))
    lit = oracle_collect(oracle, rq, rows, oracle.TIES_LITERAL)
    same_modulo_ties([r.index for r in res], [r.score for r in res], lit["index"], lit["score"])


def test_add_vectors_dim_mismatch_gpu():
    case = next(c for c in VEC_CASES if c.get("add_vectors"))
    store = VecStore(case["add_vectors"]["dim"])
    with pytest.raises(OttersError) as ei:
        store.add_vectors(case["add_vectors"]["vectors"])
    assert case["expect"]["error_contains"] in str(ei.value)
    assert store.len() == 1


def test_inv_norms_and_rows_bit_exact(oracle):
    rng = np.random.default_rng(1)
    for dim in (1, 3, 4, 7, 8, 31, 32, 33, 100, 128, 768):
        rows = rng.uniform(-1, 1, (197, dim)).astype(np.float32)
        rows[5] = 0.0  # zero norm -> inv 0 (src/vec.rs:367)
        store = VecStore(dim)
        store.add_vectors(rows[:100])
        store.add_vectors(rows[100:])  # second append lands after the first
        assert np.array_equal(store.rows(), rows)
        assert np.array_equal(store.inv_norms().view(np.uint32), oracle.inv_norms(rows).view(np.uint32))
        assert store.inv_norms()[5] == 0.0


@pytest.mark.parametrize("devices", [None, [0, 0, 0]], ids=["one_gpu", "three_shards"])
def test_single_row_appends_are_staged_and_invisible(oracle, devices):
    """VecStore::add_vector is one row per call (src/vec.rs:357-371).  Small appends are staged in pinned host memory and sent to
    the GPU 4 MB at a time, and before anything looks at the rows: len(), queries, reads, inverse norms and other kinds of
    append see every row, in order, exactly as if each append had gone to the GPU at once (option stage_appends = 0)."""
    import time
    rng = np.random.default_rng(5)
    dim, n = 96, 6000
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    q = rng.uniform(-1, 1, (3, dim)).astype(np.float32)
    store = VecStore(dim, devices=devices)
    t0 = time.perf_counter()
    for i in range(2500):
        store.add_vector(rows[i])
    per_row = (time.perf_counter() - t0) / 2500
    assert store.len() == 2500
    hits, _ = store.query(q, Metric.Cosine).take(20).collect_arrays()  # the staged rows are scored: flushed by the query
    ref = oracle.vec_query(rows[:2500], q, oracle.METRIC_COSINE, oracle.TAKE_MAX, 20, ties=oracle.TIES_CANONICAL)
    assert_bit_exact(hits, ref)
    for i in range(2500, 2600):
        store.add_vector(rows[i])
    assert np.array_equal(store.rows(2490, 110), rows[2490:2600])      # ... by a read
    store.add_vectors(rows[2600:2650])                                  # a small batch joins the staged rows
    assert np.array_equal(store.inv_norms().view(np.uint32), oracle.inv_norms(rows[:2650]).view(np.uint32))
    for i in range(2650, 2700):
        store.add_vector(rows[i])
    store.add_vectors(rows[2700:6000])                                  # a large append: the staged rows go first, the order holds
    assert store.len() == n and np.array_equal(store.rows(), rows)
    for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
        hits, _ = store.query(q, metric).take(700).collect_arrays()
        ref = oracle.vec_query(rows, q, int(metric), 0 if metric == Metric.Euclidean else 1, 700, ties=oracle.TIES_CANONICAL)
        assert_bit_exact(hits, ref)
    plain = VecStore(dim, devices=devices)
    plain.set_option("stage_appends", 0)
    t0 = time.perf_counter()
    for i in range(300):
        plain.add_vector(rows[i])
    per_row_plain = (time.perf_counter() - t0) / 300
    assert np.array_equal(plain.rows(), rows[:300])
    assert per_row < per_row_plain  # (a memcpy against a copy, a kernel and a wait per row: ~10 us against ~70 through Python)
    store.close()
    plain.close()


def test_random_fill_matches_oracle_generator(oracle):
    store = VecStore(37)
    store.append_random(300, seed=99)
    store.append_random(77, seed=99)
    want = oracle.rand_rows(0, 377, 37, 99)
    assert np.array_equal(store.rows(), want)
    assert np.array_equal(store.inv_norms().view(np.uint32), oracle.inv_norms(want).view(np.uint32))
    assert want.min() >= -1.0 and want.max() < 1.0


SHAPES = [  # (n, dim, nq)
    (1, 3, 1), (7, 4, 2), (8, 8, 1), (9, 5, 3), (63, 16, 1), (64, 33, 2), (65, 37, 5), (200, 100, 8),
    (1000, 128, 1), (1500, 768, 1), (777, 96, 9), (2100, 24, 17), (4097, 40, 4),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_d%d_q%d" % s)
@pytest.mark.parametrize("metric", [Metric.Cosine, Metric.Euclidean, Metric.DotProduct], ids=lambda m: m.name)
def test_random_parity(oracle, shape, metric):
    n, dim, nq = shape
    rng = np.random.default_rng(n * 1000 + dim + nq)
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    for k in (1, 10, 64, 65, 130, 300):
        for kind in ("take", "take_min", "take_max"):
            plan = getattr(store.query(queries, metric), kind)(k)
            rq, hits, _, stats = gpu_hits(plan)
            ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
            assert_bit_exact(hits, ref)
            assert stats["vectors_compared"] == n * nq
    # literal collector equivalence + filters
    med = float(np.median(oracle_collect(oracle, store.query(queries, metric).take(n * nq).resolve(), rows, 1)["score"])) if n * nq <= 512 else 0.1
    for cmp in (Cmp.Lt, Cmp.Gt, Cmp.Lte, Cmp.Gte):
        plan = store.query(queries, metric).filter(med, cmp).take(25)
        rq, hits, _, _ = gpu_hits(plan)
        assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
        lit = oracle_collect(oracle, rq, rows, oracle.TIES_LITERAL)
        same_modulo_ties(hits["index"], hits["score"], lit["index"], lit["score"], hits["query"], lit["query"])


def test_seq4_reduce_order(oracle):
    rng = np.random.default_rng(5)
    rows = rng.uniform(-1, 1, (500, 72)).astype(np.float32)
    q = rng.uniform(-1, 1, (2, 72)).astype(np.float32)
    store = VecStore(72)
    store.set_reduce_order(1)
    store.add_vectors(rows)
    rq, hits, _, _ = gpu_hits(store.query(q, Metric.DotProduct).take(50))
    assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL, reduce_mode=oracle.REDUCE_SEQ4))


def test_ties_and_duplicates(oracle):
    # quantised data: many exact ties, duplicates of the query itself
    rng = np.random.default_rng(7)
    rows = rng.integers(-2, 3, (900, 12)).astype(np.float32)
    queries = rng.integers(-2, 3, (3, 12)).astype(np.float32)
    rows[100] = queries[0]
    rows[700] = queries[0]
    store = VecStore(12)
    store.add_vectors(rows)
    for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
        for k in (1, 5, 40, 128):
            rq, hits, _, _ = gpu_hits(store.query(queries, metric).take(k))
            assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
            lit = oracle_collect(oracle, rq, rows, oracle.TIES_LITERAL)
            same_modulo_ties(hits["index"], hits["score"], lit["index"], lit["score"], hits["query"], lit["query"])


def test_nan_and_zero_norm(oracle):
    rows = np.array([[1, 0, 0], [np.nan, 1, 0], [0, 0, 0], [0.5, 0.5, 0], [np.inf, 0, 0]], np.float32)
    store = VecStore(3)
    store.add_vectors(rows)
    for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
        rq, hits, _, _ = gpu_hits(store.query([1.0, 0.0, 0.0], metric).take(5))
        ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
        assert_bit_exact(hits, ref)
        assert not np.isnan(hits["score"]).any()  # NaN scores are dropped, src/vec_compute.rs:237


def test_row_mask(oracle):
    rng = np.random.default_rng(11)
    n, dim = 1000, 20
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    q = rng.uniform(-1, 1, (2, dim)).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    for mask_len in (n, 517, 64, 3):  # shorter masks: missing bits keep the row (src/vec.rs:234)
        mask = rng.random(mask_len) < 0.3
        plan = store.query(q, Metric.Cosine).with_row_mask(mask).take(40)
        rq, hits, _, _ = gpu_hits(plan)
        assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
    none = store.query(q, Metric.Cosine).with_row_mask(np.zeros(n, bool)).take(40).collect()
    assert none == []


def test_per_query_mode(oracle):
    rng = np.random.default_rng(13)
    n, dim, nq = 3000, 48, 11
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    for k in (7, 100, 130, 300):  # 130 / 300: one query per pass (E >= 4) must still land in its own list slot
        res = store.query(queries, Metric.Cosine).per_query().take(k).with_path(Path.Exact).collect()
        assert len(res) == nq
        for qi in range(nq):
            ref = oracle.vec_query(rows, queries[qi], 0, 1, k, ties=oracle.TIES_CANONICAL)
            assert [r.index for r in res[qi]] == [int(i) for i in ref["index"]]
            assert np.array_equal(np.array([r.score for r in res[qi]], np.float32).view(np.uint32), ref["score"].view(np.uint32))


def test_chunk_mask_matches_meta_oracle(oracle):
    rng = np.random.default_rng(17)
    n, dim, cs = 5000, 32, 300
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    q = rng.uniform(-1, 1, (3, dim)).astype(np.float32)
    store = VecStore(dim)
    store.set_chunk_size(cs)
    store.add_vectors(rows)
    n_chunks = (n + cs - 1) // cs
    chunk_mask = rng.random(n_chunks) < 0.5
    chunk_mask[-1] = True  # the short last chunk
    row_mask = rng.random(n) < 0.7
    for metric, take in ((Metric.Cosine, 1), (Metric.Euclidean, 0)):
        plan = store.query(q, metric).with_row_mask(row_mask).filter(0.0 if metric == Metric.Cosine else 12.0, Cmp.Gt if metric == Metric.Cosine else Cmp.Lt).take(33)
        rq = plan.resolve()
        hits, _, stats = store._run(rq, chunk_mask=chunk_mask)
        ref, rstats = oracle.meta_query(rows, cs, q, rq.metric, rq.take, rq.k, rq.filter_cmp, rq.filter_thr,
                                        chunk_mask=chunk_mask, row_mask=row_mask, ties=oracle.TIES_CANONICAL)
        assert_bit_exact(hits, ref)
        for key in ("total_chunks", "pruned_chunks", "evaluated_chunks", "vectors_compared"):
            assert stats[key] == rstats[key], key


def test_base_offset_and_large_k_error():
    store = VecStore(4)
    store.set_base_offset(1_000_000_007)
    store.add_vectors(np.eye(4, dtype=np.float32))
    res = store.query([0, 0, 1, 0], Metric.DotProduct).take(1).collect()
    assert res[0].index == 1_000_000_007 + 2 and res[0].score == 1.0


def test_large_k_sort_path(oracle):
    """k > 512 (incl. collect() with no take: take_count = n_vecs, src/vec.rs:213) goes through the
    score dump + device radix sort; same canonical order as the oracle."""
    rng = np.random.default_rng(21)
    n, dim = 6000, 20
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    rows[77] = rows[5]  # an exact tie
    queries = rng.uniform(-1, 1, (3, dim)).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
        for plan in (store.query(queries[0], metric),                       # no take: k = n, take type Max even for Euclidean
                     store.query(queries, metric).take(2500),
                     store.query(queries, metric).filter(0.0, Cmp.Gt).take_min(1000),
                     store.query(queries[1], metric).with_row_mask(rng.random(n) < 0.5).take(5000)):
            rq, hits, _, stats = gpu_hits(plan)
            ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
            assert_bit_exact(hits, ref)
    res = store.query(queries, Metric.Cosine).per_query().take(700).collect()
    for qi in range(3):
        ref = oracle.vec_query(rows, queries[qi], 0, 1, 700, ties=oracle.TIES_CANONICAL)
        assert [r.index for r in res[qi]] == [int(i) for i in ref["index"]]
        assert np.array_equal(np.array([r.score for r in res[qi]], np.float32).view(np.uint32), ref["score"].view(np.uint32))


@pytest.mark.parametrize("small_sort", [1, 0], ids=["rank", "radix"])
def test_small_results_sorted_by_rank(oracle, small_sort):
    """Results of up to 16384 (row, query) pairs with k > 512 — the reference's default take on a small store (src/vec.rs:213:
    every row, sorted) — are ordered by rank in two launches (small_rank_kernel / small_place_kernel) instead of the radix sort's
    passes: merged and per query, quantised rows (long runs of equal scores: the order among them is the canonical one — lower
    row, lower query — or the reference's visit order), filters, row masks, chunk masks on a store much larger than what is
    scored.  Same bits as the oracle, and as the radix path (small_sort = 0)."""
    rng = np.random.default_rng(8)
    for n, dim, nq, quant in ((10_000, 24, 1, False), (3000, 7, 5, True), (16_384, 8, 1, True), (1000, 33, 16, False), (700, 5, 23, True)):
        rows = (rng.integers(-2, 3, (n, dim)) if quant else rng.uniform(-1, 1, (n, dim))).astype(np.float32)
        queries = (rng.integers(-2, 3, (nq, dim)) if quant else rng.uniform(-1, 1, (nq, dim))).astype(np.float32)
        queries[np.all(queries == 0, axis=1)] = 1.0
        store = VecStore(dim)
        store.set_option("small_sort", small_sort)
        store.add_vectors(rows)
        mask = rng.random(n) < 0.6
        for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
            plans = [store.query(queries, metric),                                        # the default take
                     store.query(queries, metric).take(600),
                     store.query(queries, metric).filter(0.0, Cmp.Gte).take_min(2000),
                     store.query(queries, metric).with_row_mask(mask).take(n)]
            for plan in plans:
                if plan.resolve().k <= 512:
                    continue
                rq, hits, _, _ = gpu_hits(plan)
                ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
                assert_bit_exact(hits, ref)
            if n * nq <= 16_384 or True:
                res, counts = store.query(queries, metric).per_query().take(900).collect_arrays()
                o = 0
                for qi in range(nq):
                    ref = oracle.vec_query(rows, queries[qi], int(metric), 0 if metric == Metric.Euclidean else 1, 900, ties=oracle.TIES_CANONICAL)
                    g = res[o:o + counts[qi]]
                    assert np.array_equal(g["index"], ref["index"]) and np.array_equal(g["score"].view(np.uint32), ref["score"].view(np.uint32)) and np.all(g["query"] == qi)
                    o += counts[qi]
        # the reference's tie order through the same path (k + 1 candidates in visit order, the flat fill pass)
        store.set_tie_order("reference")
        for k in (513, 1500):
            if k > n * nq:
                continue
            got, _ = store.query(queries, Metric.DotProduct).take(k).collect_arrays()
            lit = oracle.vec_query(rows, queries, oracle.METRIC_DOT, oracle.TAKE_MAX, k, ties=oracle.TIES_LITERAL)
            assert np.array_equal(got["score"].view(np.uint32), lit["score"].view(np.uint32))
            assert sorted(zip(got["index"].tolist(), got["query"].tolist())) == sorted(zip(lit["index"].tolist(), lit["query"].tolist()))
        store.close()
    # a large store of which a chunk mask leaves little: the pairs scored decide, not the store's size
    big = VecStore(16)
    big.set_option("small_sort", small_sort)
    big.set_chunk_size(1000)
    big.append_random(300_000, 3)
    rows = oracle.rand_rows(0, 300_000, 16, 3)
    cm = np.zeros(300, bool)
    cm[[7, 100, 299]] = True
    q = rng.uniform(-1, 1, (2, 16)).astype(np.float32)
    rq = big.query(q, Metric.Cosine).take(5000).resolve()
    hits, _, st = big._run(rq, chunk_mask=cm)
    sel = np.concatenate([np.arange(7000, 8000), np.arange(100_000, 101_000), np.arange(299_000, 300_000)])
    ref = oracle.vec_query(rows[sel], q, oracle.METRIC_COSINE, oracle.TAKE_MAX, 5000, ties=oracle.TIES_CANONICAL)
    assert np.array_equal(hits["index"], sel[ref["index"]]) and np.array_equal(hits["score"].view(np.uint32), ref["score"].view(np.uint32))
    assert np.array_equal(hits["query"], ref["query"]) and st["vectors_compared"] == 6000
    big.close()


@pytest.mark.parametrize("order", ["random", "best_last", "best_first", "quantised"])
@pytest.mark.parametrize("pre", [1, 0])
def test_large_k_two_phase_gate(oracle, order, pre):
    """k > 512 on a store large enough for the two-phase large-k path (option large_k_pre: a prefix of the rows is scored and
    sorted first; of the rest only pairs that reach its k-th best are listed): rows in random order, with the best rows LAST
    (the bound prunes nothing), FIRST (it prunes everything else) and on a quantised corpus whose k-th place sits inside a
    long run of equal scores — merged and per query, filters, row and chunk masks, all metrics: the oracle's order and bits."""
    rng = np.random.default_rng(33)
    n, dim, cs = 90_000, 16, 1000
    if order == "quantised":
        rows = rng.integers(-2, 3, (n, dim)).astype(np.float32)
        queries = rng.integers(-2, 3, (3, dim)).astype(np.float32)
        queries[np.all(queries == 0, axis=1)] = 1.0
    else:
        rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        queries = rng.uniform(-1, 1, (3, dim)).astype(np.float32)
        if order != "random":  # sort by the first query's dot product
            key = rows @ queries[0]
            rows = rows[np.argsort(key if order == "best_last" else -key, kind="stable")]
    store = VecStore(dim)
    store.set_chunk_size(cs)
    store.set_option("force_fallback", 0 if pre else 8)  # bit 8: the sort path lists every pair (no prefix gate)
    store.add_vectors(rows)
    row_mask = rng.random(n) < 0.8
    chunk_mask = rng.random((n + cs - 1) // cs) < 0.8
    for metric in (Metric.DotProduct, Metric.Cosine, Metric.Euclidean):
        for k in (600, 3000):
            for perq in (False, True):
                for variant in range(3):
                    plan = store.query(queries, metric)
                    plan = plan.take_min(k) if (variant == 1) else plan.take(k)
                    if perq:
                        plan = plan.per_query()
                    cm = None
                    if variant == 2:
                        plan = plan.with_row_mask(row_mask).filter(0.0 if metric != Metric.Euclidean else float(dim) * 0.6, Cmp.Gt)
                        cm = chunk_mask
                    rq = plan.resolve()
                    hits, counts, _ = store._run(rq, chunk_mask=cm)
                    if perq:
                        per = [oracle.meta_query(rows, cs, queries[q:q + 1], rq.metric, rq.take, rq.k, rq.filter_cmp, rq.filter_thr, chunk_mask=cm,
                                                 row_mask=rq.row_mask, ties=oracle.TIES_CANONICAL)[0] for q in range(3)]
                        for q, h in enumerate(per):
                            h["query"] = q
                        ref = np.concatenate(per)
                        assert [int(c) for c in counts] == [len(h) for h in per]
                    else:
                        ref = oracle.meta_query(rows, cs, queries, rq.metric, rq.take, rq.k, rq.filter_cmp, rq.filter_thr, chunk_mask=cm,
                                                row_mask=rq.row_mask, ties=oracle.TIES_CANONICAL)[0]
                    ctx = (order, pre, metric.name, k, perq, variant)
                    assert hits.shape == ref.shape, ctx
                    assert np.array_equal(hits["index"], ref["index"]) and np.array_equal(hits["query"], ref["query"]), ctx
                    assert np.array_equal(hits["score"].view(np.uint32), ref["score"].view(np.uint32)), ctx


def test_results_of_hundreds_of_thousands_of_hits(oracle):
    """Results large enough (>= 256k hits) to come to the host in pinned pieces, straight into the caller's buffer: the
    reference's default take (every row, src/vec.rs:213) on 300k rows, per-query lists of all rows for two queries, a merged
    take(400k) over two queries, with a filter and a row mask — order, owners and score bits are the oracle's."""
    rng = np.random.default_rng(77)
    n, dim = 300_000, 16
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    rows[1000:1010] = rows[5]  # exact ties
    queries = rng.uniform(-1, 1, (2, dim)).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    mask = rng.random(n) < 0.95
    plans = [store.query(queries[0], Metric.Cosine),
             store.query(queries, Metric.DotProduct).take(400_000),
             store.query(queries, Metric.Euclidean).take_min(400_000).with_row_mask(mask),
             store.query(queries[1], Metric.Cosine).filter(-0.9, Cmp.Gt)]
    for plan in plans:
        rq, hits, _, stats = gpu_hits(plan)
        ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
        assert hits.size >= 256 * 1024
        assert_bit_exact(hits, ref)
    got, counts = store.query(queries, Metric.Cosine).per_query().collect_arrays()
    assert [int(c) for c in counts] == [n, n]
    for qi in range(2):
        ref = oracle.vec_query(rows, queries[qi], 0, 1, n, ties=oracle.TIES_CANONICAL)
        seg = got[qi * n:(qi + 1) * n]
        assert np.array_equal(seg["index"], ref["index"]) and np.all(seg["query"] == qi)
        assert np.array_equal(seg["score"].view(np.uint32), ref["score"].view(np.uint32))


def test_concurrent_queries_from_threads(oracle):
    """ott_query is re-entrant on one store (overlapping calls run on worker contexts): concurrent host threads get correct, independent results"""
    import threading
    rng = np.random.default_rng(31)
    rows = rng.uniform(-1, 1, (20000, 32)).astype(np.float32)
    store = VecStore(32)
    store.add_vectors(rows)
    qs = rng.uniform(-1, 1, (8, 32)).astype(np.float32)
    want = [oracle.vec_query(rows, q, 0, 1, 10, ties=oracle.TIES_CANONICAL) for q in qs]
    errs = []

    def work(i):
        try:
            for _ in range(20):
                res = store.query(qs[i], Metric.Cosine).take(10).collect()
                assert [r.index for r in res] == [int(x) for x in want[i]["index"]]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_collect_arrays_matches_collect():
    rng = np.random.default_rng(41)
    n, dim, nq = 4000, 40, 6
    store = VecStore(dim)
    store.add_vectors(rng.uniform(-1, 1, (n, dim)).astype(np.float32))
    queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    merged = store.query(queries, Metric.Cosine).take(25)
    hits, counts = merged.collect_arrays()
    objs = merged.collect()
    assert [int(i) for i in hits["index"]] == [r.index for r in objs]
    assert [float(x) for x in hits["score"]] == [r.score for r in objs]
    assert sum(counts) == len(objs) == 25
    perq = store.query(queries, Metric.Cosine).per_query().take(9)
    hits, counts = perq.collect_arrays()
    lists = perq.collect()
    assert counts == [9] * nq and len(hits) == 9 * nq
    o = 0
    for qi in range(nq):
        assert np.all(hits["query"][o:o + 9] == qi)
        assert [int(i) for i in hits["index"][o:o + 9]] == [r.index for r in lists[qi]]
        o += 9


def test_concurrent_queries_from_host_threads(oracle):
    """ott_query on one store from several host threads (ctypes drops the GIL): overlapping calls run on worker
    contexts and every one of them must return what it returns alone."""
    import threading
    rng = np.random.default_rng(43)
    n, dim = 20000, 64
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    queries = rng.uniform(-1, 1, (8, dim)).astype(np.float32)
    expect = [oracle.vec_query(rows, queries[i], 0, 1, 10, ties=oracle.TIES_CANONICAL) for i in range(8)]
    errors = []

    def worker(i):
        try:
            for _ in range(20):
                hits, _ = store.query(queries[i], Metric.Cosine).take(10).collect_arrays()
                if [int(x) for x in hits["index"]] != [int(x) for x in expect[i]["index"]]:
                    errors.append((i, "index"))
                if not np.array_equal(hits["score"].view(np.uint32), expect[i]["score"].view(np.uint32)):
                    errors.append((i, "score"))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []


def test_errors_through_the_c_abi():
    """Misuse is an error code + message, never a crash: NULL arguments, unknown enum values, short output buffers,
    an allocation the GPU cannot satisfy (the store stays usable afterwards)."""
    import ctypes as C
    from otters_amd import _native as N
    L = N.lib()
    assert L.ott_store_create(0, 0, C.byref(C.c_void_p())) != 0 and b"dim" in L.ott_last_error()
    assert L.ott_store_create(8, 0, None) != 0
    assert L.ott_store_append(None, None, 1) != 0
    store = VecStore(8)
    store.add_vectors(np.ones((100, 8), np.float32))
    h = store._handle()
    out = np.zeros(16, dtype=N.HIT_DTYPE)
    q = np.ones(8, np.float32)
    d = N.QueryDesc()
    d.queries, d.nq, d.metric, d.take, d.k = q.ctypes.data, 1, 0, 1, 10
    n_out = C.c_uint64(0)
    assert L.ott_query(h, C.byref(d), N.ptr(out), 16, C.byref(n_out), None, None) == 0 and n_out.value == 10
    assert L.ott_query(h, C.byref(d), N.ptr(out), 4, C.byref(n_out), None, None) != 0       # capacity < min(k, rows)
    assert b"capacity" in L.ott_last_error()
    assert L.ott_query(h, None, N.ptr(out), 16, C.byref(n_out), None, None) != 0
    assert L.ott_query(h, C.byref(d), None, 16, C.byref(n_out), None, None) != 0
    for field, bad in (("metric", 7), ("take", 5), ("filter_cmp", 9), ("mode", 3), ("path", 9)):
        d2 = N.QueryDesc()
        d2.queries, d2.nq, d2.metric, d2.take, d2.k = q.ctypes.data, 1, 0, 1, 10
        setattr(d2, field, bad)
        assert L.ott_query(h, C.byref(d2), N.ptr(out), 16, C.byref(n_out), None, None) != 0, field
    d3 = N.QueryDesc()
    d3.queries, d3.nq, d3.metric, d3.take, d3.k = None, 1, 0, 1, 10
    assert L.ott_query(h, C.byref(d3), N.ptr(out), 16, C.byref(n_out), None, None) != 0
    # an impossible reservation fails cleanly and leaves the rows in place
    with pytest.raises(OttersError):
        store.reserve(10**13)
    assert store.len() == 100
    assert [r.index for r in store.query(q, Metric.DotProduct).take(3).collect()] == [0, 1, 2]
    # ... and one that reaches hipMalloc (200M x 768 f32 = 614 GB): OTT_ERR_OOM, the rows stay, and the failure is not seen a
    # second time by the launch checks of the queries and appends that follow
    wide = VecStore(768)
    wide.add_vectors(np.ones((10, 768), np.float32))
    with pytest.raises(OttersError) as ei:
        wide.reserve(200_000_000)
    assert ei.value.status == -3 and "hipMalloc" in str(ei.value), (ei.value.status, str(ei.value))
    assert wide.len() == 10
    assert [r.index for r in wide.query(np.ones(768, np.float32), Metric.Cosine).take(3).collect()] == [0, 1, 2]
    wide.add_vectors(np.full((5, 768), 2.0, np.float32))
    assert [r.index for r in wide.query(np.ones(768, np.float32), Metric.DotProduct).take(2).collect()] == [10, 11]


@pytest.mark.parametrize("small", ["0", "2"])
def test_small_grid_kernel_variant_matches_streaming_kernel(oracle, small, monkeypatch):
    """Single queries on small stores run a small-store variant of the exact kernel — rows8 (eight lanes per row, lane l
    owning accumulator chain l of the reference's f32x8, src/vec_compute.rs:9-22; the default); OTT_EXACT_SMALL forces the
    streaming kernel (0) or rows8 (2), so both are held to the oracle on the same inputs (round 2's one-wave LDS-DMA variant,
    1, was retired in round 5) (ragged tiles, dims that are not multiples of 4 / 8 / 32, all metrics, filters, masks, chunk runs)."""
    monkeypatch.setenv("OTT_EXACT_SMALL", small)
    rng = np.random.default_rng(77)
    for n, dim in ((1, 3), (63, 7), (64, 8), (65, 33), (700, 100), (3001, 768), (9000, 130), (20000, 1030)):
        rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        store = VecStore(dim)
        store.set_chunk_size(256)
        store.add_vectors(rows)
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        mask = rng.random(n) < 0.7
        for metric, take in ((Metric.Cosine, 1), (Metric.Euclidean, 0), (Metric.DotProduct, 1)):
            for k in (1, 10, 100):
                plan = store.query(q, metric).take(k).with_path(Path.Exact)
                rq, hits, _, _ = gpu_hits(plan)
                assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
            plan = store.query(q, metric).with_row_mask(mask).take(20).with_path(Path.Exact)
            rq, hits, _, _ = gpu_hits(plan)
            assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
            plan = store.query(q, metric).filter(0.0, Cmp.Gt).take(7).with_path(Path.Exact)
            rq, hits, _, _ = gpu_hits(plan)
            assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
        if n >= 700:  # a zonemap-style chunk mask: every other 256-row chunk, so tiles map to several runs
            n_chunks = (n + 255) // 256
            cm = (np.arange(n_chunks) % 2) == 0
            keep = np.repeat(cm, 256)[:n]
            rq = store.query(q, Metric.Cosine).take(10).with_path(Path.Exact).resolve()
            h, _, st = store._run(rq, chunk_mask=cm)
            ref = oracle.vec_query(rows, q, oracle.METRIC_COSINE, oracle.TAKE_MAX, 10, row_mask=keep, ties=oracle.TIES_CANONICAL)
            assert np.array_equal(h["index"], ref["index"]) and np.array_equal(h["score"].view(np.uint32), ref["score"].view(np.uint32))
        # small batches (rows8 takes up to 8 queries per pass, up to 16 per call; the streaming kernel 4 per pass): merged and per query
        for nq in (2, 3, 5, 8, 11, 16):
            Q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
            for metric, take in ((Metric.Cosine, 1), (Metric.Euclidean, 0)):
                for k in (3, 70):
                    plan = store.query(Q, metric).take(k).with_path(Path.Exact)
                    rq, hits, _, _ = gpu_hits(plan)
                    assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
                    got, counts = store.query(Q, metric).take(k).per_query().with_path(Path.Exact).collect_arrays()
                    o = 0
                    for qi in range(nq):
                        ref = oracle.vec_query(rows, Q[qi], int(metric), take, k, ties=oracle.TIES_CANONICAL)
                        g = got[o:o + counts[qi]]
                        assert np.array_equal(g["index"], ref["index"]) and np.array_equal(g["score"].view(np.uint32), ref["score"].view(np.uint32)), (n, dim, nq, metric, k, qi)
                        assert np.all(g["query"] == qi)
                        o += counts[qi]
            plan = store.query(Q, Metric.DotProduct).with_row_mask(mask).filter(0.0, Cmp.Gt).take(9).with_path(Path.Exact)
            rq, hits, _, _ = gpu_hits(plan)
            assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
        # the other horizontal-sum order of wide::f32x8::reduce_add
        store.set_reduce_order(1)
        rq = store.query(q, Metric.DotProduct).take(10).with_path(Path.Exact).resolve()
        h, _, _ = store._run(rq)
        ref = oracle.vec_query(rows, q, oracle.METRIC_DOT, oracle.TAKE_MAX, 10, reduce_mode=oracle.REDUCE_SEQ4, ties=oracle.TIES_CANONICAL)
        assert np.array_equal(h["index"], ref["index"]) and np.array_equal(h["score"].view(np.uint32), ref["score"].view(np.uint32))


def test_concurrent_batches_from_host_threads(oracle):
    """Batch queries (the cascade: hi plane built on first use, per-context scratch, shared back-off state) from several host
    threads at once on one store: every call must return what it returns alone, from the very first (racing) calls on."""
    import threading
    rng = np.random.default_rng(47)
    n, dim, nq = 30000, 96, 12
    rows = rng.normal(0, 1, (n, dim)).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    batches = [rng.normal(0, 1, (nq, dim)).astype(np.float32) for _ in range(6)]
    expect = [oracle.vec_query(rows, b, int(Metric.Cosine), 1, 7, ties=oracle.TIES_CANONICAL) for b in batches]
    errors = []

    def worker(i):
        try:
            for _ in range(10):
                hits, _ = store.query(batches[i], Metric.Cosine).take(7).collect_arrays()
                if store.last_stats is None:
                    errors.append((i, "stats"))
                if not (np.array_equal(hits["index"], expect[i]["index"]) and
                        np.array_equal(hits["score"].view(np.uint32), expect[i]["score"].view(np.uint32))):
                    errors.append((i, "hits"))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []
    hits, _ = store.query(batches[0], Metric.Cosine).take(7).collect_arrays()
    assert store.last_stats["path_used"] == 2  # these batches do take the batch path


@pytest.mark.parametrize("n", [4095, 4096, 4097, 8193, 12289, 70001])
def test_large_k_radix_sort_at_tile_boundaries(oracle, n):
    """take beyond the register lists (k > 512, here: every pair) goes through the score dump and the device radix sort (one
    kernel per digit with decoupled look-back over tiles of 4096 pairs): pair counts just below / at / above one tile and
    several tiles, heavily tied scores (quantised rows: the row and query digits decide most of the order), one and several
    queries (the query-id digit passes), merged and grouped by query, with and without a filter (fewer pairs than rows) — the
    whole list against the oracle's canonical order, and the reference's visit order (tie_order) against its literal collector."""
    rng = np.random.default_rng(n)
    dim = 4
    rows = rng.integers(-2, 3, (n, dim)).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    for nq in (1, 3):
        q = rng.integers(-2, 3, (nq, dim)).astype(np.float32)
        q[np.all(q == 0, axis=1)] = 1.0
        for metric, take in ((Metric.DotProduct, 1), (Metric.Euclidean, 0)):
            plan = store.query(q, metric).with_path(Path.Exact)
            plan = plan.take_max(n * nq) if take else plan.take_min(n * nq)
            got, _ = plan.collect_arrays()
            ref = oracle.vec_query(rows, q, int(metric), take, n * nq, ties=oracle.TIES_CANONICAL)
            assert got.size == n * nq
            assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["query"], ref["query"]), (n, nq, metric)
            assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
            # a filter: the dump holds fewer pairs than (rows x queries); k well above 512 but below the count
            plan = store.query(q, metric).filter(1.0, Cmp.Gte).with_path(Path.Exact)
            got, _ = (plan.take_max(3000) if take else plan.take_min(3000)).collect_arrays()
            ref = oracle.vec_query(rows, q, int(metric), take, 3000, oracle.CMP_GTE, 1.0, ties=oracle.TIES_CANONICAL)
            assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["query"], ref["query"]), (n, nq, metric, "filtered")
            # grouped by query
            got, counts = store.query(q, metric).with_path(Path.Exact).per_query().take(700).collect_arrays()
            assert counts == [700] * nq
            for qi in range(nq):
                ref = oracle.vec_query(rows, q[qi], int(metric), 1 if metric != Metric.Euclidean else 0, 700, ties=oracle.TIES_CANONICAL)
                g = got[qi * 700:(qi + 1) * 700]
                assert np.array_equal(g["index"], ref["index"]) and np.all(g["query"] == qi), (n, nq, metric, qi)
    # the reference's visit order through the same sort (three-part pass plan: row & 7, query, then block and score)
    store.set_tie_order("reference")
    q = rng.integers(-2, 3, (3, dim)).astype(np.float32)
    q[np.all(q == 0, axis=1)] = 1.0
    got, _ = store.query(q, Metric.DotProduct).take(2000).collect_arrays()
    lit = oracle.vec_query(rows, q, oracle.METRIC_DOT, oracle.TAKE_MAX, 2000, ties=oracle.TIES_LITERAL)
    assert np.array_equal(got["score"].view(np.uint32), lit["score"].view(np.uint32))
    assert sorted(zip(got["index"].tolist(), got["query"].tolist())) == sorted(zip(lit["index"].tolist(), lit["query"].tolist()))
    store.close()


@pytest.mark.parametrize("tie", ["canonical", "reference"])
def test_sort_path_in_row_slices_with_carried_gates(oracle, tie):
    """Round 5: beyond 2^29 (row, query) pairs the sort path (k > 512, the default take) cuts the rows into slices, merges the
    slices' sorted results on the host and carries the running k-th best into the next slices as their gate (run_large_k).
    force_fallback bit 64 cuts at 2^14 pairs, so stores of tens of thousands of rows run dozens of slices: the result must equal
    the one-slice result bit for bit — merged and per query, k from just above the lists to every pair, filters, masks, chunk
    masks, both tie orders — and the oracle's where the oracle is quick."""
    rng = np.random.default_rng(12)
    for n, dim, nq in ((20_011, 24, 1), (30_000, 16, 3), (9_000, 40, 9), (70_000, 8, 2)):
        rows = (rng.integers(-3, 4, (n, dim)) if tie == "reference" else rng.uniform(-1, 1, (n, dim))).astype(np.float32)
        qs = (rng.integers(-2, 3, (nq, dim)) if tie == "reference" else rng.uniform(-1, 1, (nq, dim))).astype(np.float32)
        qs[np.all(qs == 0, axis=1)] = 1.0
        one, cut = VecStore(dim), VecStore(dim)
        for s in (one, cut):
            s.set_chunk_size(1000)
            s.set_tie_order(tie)
            s.add_vectors(rows)
        cut.set_option("force_fallback", 64)
        mask = rng.random(n) < 0.6
        cmask = rng.random((n + 999) // 1000) < 0.7
        for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
            for k in (513, 2000, n // 2, None):
                for variant in ("plain", "filter", "row_mask", "chunk_mask", "perq"):
                    def plan(s):
                        p = s.query(qs, metric)
                        if variant == "filter":
                            p = p.filter(0.0 if metric != Metric.Euclidean else float(dim), Cmp.Gt)
                        if variant == "row_mask":
                            p = p.with_row_mask(mask)
                        if variant == "perq":
                            p = p.per_query()
                        if k is not None:
                            p = p.take(k) if metric != Metric.Euclidean else p.take_min(k)
                        p = p.with_path(Path.Exact)
                        rq = p.resolve()
                        return s._run(rq, chunk_mask=cmask if variant == "chunk_mask" else None)[:2]
                    a, ca = plan(one)
                    b, cb = plan(cut)
                    where = (tie, n, dim, nq, metric, k, variant)
                    assert a.shape == b.shape and ca == cb, where
                    assert np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)), where
                    if tie == "canonical":
                        assert np.array_equal(a["index"], b["index"]) and np.array_equal(a["query"], b["query"]), where
                    else:  # the reference's outcome is a set at equal scores
                        assert sorted(zip(a["index"].tolist(), a["query"].tolist())) == sorted(zip(b["index"].tolist(), b["query"].tolist())), where
        if tie == "canonical":
            ref = oracle.vec_query(rows, qs, oracle.METRIC_COSINE, oracle.TAKE_MAX, 700, ties=oracle.TIES_CANONICAL)
            got, _ = cut.query(qs, Metric.Cosine).take(700).with_path(Path.Exact).collect_arrays()
            assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
        one.close()
        cut.close()
