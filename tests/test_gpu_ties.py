"""GPU: store option tie_order = reference — at exact score ties the library keeps exactly the (row, query) pairs the
reference's TopKCollector keeps (src/vec_compute.rs:236-277, visit order of src/vec.rs:222-303), checked against the oracle's
literal restatement (OTTO_TIES_LITERAL) on quantised data where nearly every cut at take(k) runs through a group of equal
scores.  Tightened form of helpers.same_modulo_ties: the score sequences are identical AND the (index, query) sets are
identical — including the group cut by k.  VecStore: one collector over the store; MetaStore: one collector per chunk, then
concat-sort-truncate (src/meta.rs:678-709).  The default (canonical) order is untouched: the rest of the suite runs on it."""
import os

import numpy as np
import pytest

from otters_amd import Cmp, Column, DataType, MetaStore, Metric, Path, VecStore, col

pytestmark = pytest.mark.gpu


def same_sets(got, lit, where):
    assert got.size == lit.size, (where, got.size, lit.size)
    assert np.array_equal(got["score"].view(np.uint32), lit["score"].view(np.uint32)), (where, "score sequence")
    a = sorted(zip(got["index"].tolist(), got["query"].tolist()))
    b = sorted(zip(lit["index"].tolist(), lit["query"].tolist()))
    assert a == b, (where, [x for x in a if x not in b], [x for x in b if x not in a])


def quantised(rng, n, dim, levels):
    return rng.integers(-levels, levels + 1, (n, dim)).astype(np.float32)


@pytest.mark.parametrize("seed", range(int(os.environ.get("OTT_TIE_SEEDS", "12"))))  # OTT_TIE_SEEDS=300 for a soak
def test_vecstore_reference_tie_order_equals_the_literal_collector(oracle, seed):
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([7, 64, 65, 203, 1000, 4099, 20011]))
    dim = int(rng.choice([2, 3, 8, 12, 33]))
    nq = int(rng.choice([1, 1, 2, 3, 6, 9]))
    rows = quantised(rng, n, dim, int(rng.choice([1, 2, 3])))
    queries = quantised(rng, nq, dim, 2)
    queries[np.all(queries == 0, axis=1)] = 1.0
    store = VecStore(dim)
    store.set_tie_order("reference")
    store.add_vectors(rows)
    mask = (rng.random(n) < 0.85) if seed % 3 == 0 else None
    for metric, take in ((Metric.DotProduct, 1), (Metric.Euclidean, 0), (Metric.Cosine, 1), (Metric.DotProduct, 0)):
        for k in (1, 3, 10, 20, 64, 65, 130, 600, n * nq):
            if k > n * nq:
                continue
            for filt in (None, (float(rng.integers(-3, 4)), Cmp.Gte), (float(rng.integers(-3, 4)), Cmp.Lt)):
                for path in ((Path.Exact, Path.Auto) if nq > 1 else (Path.Exact,)):
                    plan = store.query(queries, metric)
                    if mask is not None:
                        plan = plan.with_row_mask(mask)
                    if filt:
                        plan = plan.filter(*filt)
                    plan = (plan.take_max(k) if take else plan.take_min(k)).with_path(path)
                    got, _ = plan.collect_arrays()
                    lit = oracle.vec_query(rows, queries, int(metric), take, k, int(filt[1]) if filt else 0, filt[0] if filt else 0.0,
                                           row_mask=mask, ties=oracle.TIES_LITERAL)
                    same_sets(got, lit, (seed, n, dim, nq, metric, take, k, filt, path))
    store.close()


def test_vecstore_reference_ties_per_query_and_batch_cascade(oracle):
    """PER_QUERY (an extension: every query gets the result of a single-query collect of its own) and batches that AUTO
    sends through the matrix-core cascade (ties are exactly what its certification cannot decide: they end on the exact
    path, in the reference's order)."""
    rng = np.random.default_rng(77)
    n, dim, nq = 30_000, 16, 24
    rows = quantised(rng, n, dim, 2)
    queries = quantised(rng, nq, dim, 2)
    queries[np.all(queries == 0, axis=1)] = 1.0
    store = VecStore(dim)
    store.set_tie_order("reference")
    store.add_vectors(rows)
    for k in (5, 40, 100):
        got, counts = store.query(queries, Metric.DotProduct).take(k).per_query().collect_arrays()
        o = 0
        for qi in range(nq):
            lit = oracle.vec_query(rows, queries[qi], oracle.METRIC_DOT, oracle.TAKE_MAX, k, ties=oracle.TIES_LITERAL)
            g = got[o:o + counts[qi]].copy()
            assert np.all(g["query"] == qi)
            g["query"] = 0
            same_sets(g, lit, ("perq", k, qi))
            o += counts[qi]
        m, _ = store.query(queries, Metric.Cosine).take(k).collect_arrays()
        lit = oracle.vec_query(rows, queries, oracle.METRIC_COSINE, oracle.TAKE_MAX, k, ties=oracle.TIES_LITERAL)
        same_sets(m, lit, ("merged batch", k))
    # back to the canonical order on the same store: bit-identical to the oracle's canonical collector again
    store.set_tie_order("canonical")
    got, _ = store.query(queries, Metric.DotProduct).take(50).collect_arrays()
    ref = oracle.vec_query(rows, queries, oracle.METRIC_DOT, oracle.TAKE_MAX, 50, ties=oracle.TIES_CANONICAL)
    assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["query"], ref["query"])
    store.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("OTT_TIE_META_SEEDS", "8"))))  # OTT_TIE_META_SEEDS=200 for a soak
def test_metastore_reference_tie_order_equals_per_chunk_collectors(oracle, seed):
    rng = np.random.default_rng(8000 + seed)
    cs = int(rng.choice([8, 16, 64, 256]))
    n = int(rng.choice([100, 1000, 5003]))
    dim = int(rng.choice([2, 4, 9]))
    nq = int(rng.choice([1, 2, 4]))
    rows = quantised(rng, n, dim, int(rng.choice([1, 2])))
    queries = quantised(rng, nq, dim, 2)
    queries[np.all(queries == 0, axis=1)] = 1.0
    bucket = ((np.arange(n) // cs) % 3).astype(np.int32)
    meta = MetaStore.from_columns([Column.from_numpy("bucket", DataType.Int32, bucket)]).with_vectors(rows).with_chunk_size(cs).build()
    meta.set_tie_order("reference")
    n_chunks = (n + cs - 1) // cs
    for metric, take in ((Metric.DotProduct, 1), (Metric.Euclidean, 0), (Metric.Cosine, 1)):
        for k in (1, 4, 10, 30, 100):
            for with_filter in (False, True):
                plan = meta.query_batch(queries, metric) if nq > 1 else meta.query(queries[0], metric)
                cmask = None
                if with_filter:
                    plan = plan.meta_filter(col("bucket").neq(1))
                    cmask = (np.arange(n_chunks) % 3) != 1
                res = plan.take(k).collect()
                lit, _ = oracle.meta_query(rows, cs, queries, int(metric), take, k, chunk_mask=cmask, ties=oracle.TIES_LITERAL)
                assert np.array_equal(np.array(res.scores, np.float32).view(np.uint32), lit["score"].view(np.uint32)), (seed, metric, k, with_filter)
                # the reference discards the query id (src/meta.rs:693-697): rows as a multiset
                assert sorted(res.indices) == sorted(lit["index"].tolist()), (seed, cs, n, nq, metric, k, with_filter)


@pytest.mark.parametrize("devices", [None, [0, 0, 0]], ids=["one_store", "three_shards"])
@pytest.mark.parametrize("cs", [3, 5, 100, 1000, 1021])
def test_metastore_tie_order_for_chunk_sizes_that_are_not_multiples_of_8(oracle, cs, devices):
    """Round 5: the reference accepts any chunk_size >= 1 (src/meta.rs:86-89), and every chunk is a VecStore of its own
    (src/meta_compute.rs:153-192): its collector visits ITS rows in blocks of eight counted from ITS first row, its remainder
    rows last.  tie_order 2 used to need chunk sizes that are multiples of 8 (chunk-local blocks = the store's); now the
    per-chunk re-queries rank by the chunk's own blocks (CoreOpts::tie_off), for 3-, 5-, 100-, 1000- and 1021-row chunks, on one
    store and over three shards (whose boundaries are multiples of lcm(chunk size, 8) rows), against the oracle's literal
    per-chunk collectors."""
    rng = np.random.default_rng(9100 + cs)
    n = {3: 500, 5: 803, 100: 3210, 1000: 12_345, 1021: 9000}[cs]
    for dim, nq, levels in ((3, 1, 1), (4, 3, 2), (9, 2, 1)):
        rows = quantised(rng, n, dim, levels)
        queries = quantised(rng, nq, dim, 2)
        queries[np.all(queries == 0, axis=1)] = 1.0
        bucket = ((np.arange(n) // cs) % 3).astype(np.int32)
        meta = (MetaStore.from_columns([Column.from_numpy("bucket", DataType.Int32, bucket)], devices=devices)
                .with_vectors(rows).with_chunk_size(cs).build())
        meta.set_tie_order("reference")
        n_chunks = (n + cs - 1) // cs
        for metric, take in ((Metric.DotProduct, 1), (Metric.Euclidean, 0), (Metric.Cosine, 1)):
            for k in (1, 2, 7, 10, 33, 100, 600):
                for with_filter in (False, True):
                    plan = meta.query_batch(queries, metric) if nq > 1 else meta.query(queries[0], metric)
                    cmask = None
                    if with_filter:
                        plan = plan.meta_filter(col("bucket").neq(1))
                        cmask = (np.arange(n_chunks) % 3) != 1
                    res = plan.take(k).collect()
                    lit, _ = oracle.meta_query(rows, cs, queries, int(metric), take, k, chunk_mask=cmask, ties=oracle.TIES_LITERAL)
                    where = (cs, devices, dim, nq, metric, k, with_filter)
                    assert np.array_equal(np.array(res.scores, np.float32).view(np.uint32), lit["score"].view(np.uint32)), where
                    assert sorted(res.indices) == sorted(lit["index"].tolist()), where
