"""GPU: ONE store over several GPUs of this process (ott_store_create_multi, VecStore(devices=[...])) — the reference's own
single-process fan-out and merge (src/meta.rs:678-709, src/vec.rs:387) with shards in place of rayon tasks.

The GPU box has one MI355X, so the device lists repeat ordinal 0 ([0,0,0,0], [0]*8, ...): real shards, per-shard streams and
host threads, the sliced masks and columns, the exchange into the merging shard's buffer, merge_hits_kernel, the row moves
between shards.  Bar: the hits (index, score bits, query) equal those of ONE single-GPU store holding the same rows, bit for
bit — three metrics, k <= 512 and beyond, the default take, PER_QUERY, score filters, chunk masks, host and device row masks,
both reference tie orders, the 1024-query cascade — and the oracle's where it is cheap.  The RCCL transport (one communicator
per device, grouped ncclAllGather) runs for real on the one-device list [0].

Every test here runs in every EXCHANGE MODE the one GPU allows (fixture `exchange_mode`; round 5):
  local      shards of a repeated ordinal share the merging GPU: blocks are written straight into its receive buffer
  remote     OTT_MULTI_FAKE_DISTINCT=1: every shard counts as a device of its own, so the exchange, the row moves, the tie
             re-queries and device appends take the branch of DISTINCT GPUs — own send buffer, hipMemcpyPeerAsync, the event
             the merging stream waits on — although the copy stays on GPU 0 (peer copies forced: OTT_MULTI_TRANSPORT=1)
  fake_rccl  the same, and the grouped ncclAllGather branch over tests/fake_rccl (N ranks on one device behind the nccl*
             names): ncclCommInitAll / ncclGroupStart / G x ncclAllGather / ncclGroupEnd with G = 3 / 4 / 8, recv = block x G
The RCCL binding is loaded once per process, so fake_rccl runs in a child process (tests/test_gpu_multi_modes.py starts this
file with OTT_TEST_MULTI_MODE=fake_rccl; also under the device-affinity audit build of the library)."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import FAKE_RCCL, build_meta_case, check_expect, load, meta_plan_from_case, multi_mode, plan_from_case
from otters_amd import Cmp, Column, DataType, MetaStore, Metric, OttersError, Path, VecStore, col
from otters_amd import _native as N

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("exchange_mode")]

DEVS = [[0, 0, 0, 0], [0] * 8]


def expected_transport():
    return "rccl" if multi_mode() == "fake_rccl" else "peer"


def same_hits(a, b, where=None):
    assert a.shape == b.shape, (where, a.shape, b.shape)
    assert np.array_equal(a["index"], b["index"]), (where, a[:8], b[:8])
    assert np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)), (where, a[:8], b[:8])
    assert np.array_equal(a["query"], b["query"]), (where, a[:8], b[:8])


def pair(dim, devs, n, seed, reserve=True, chunk_size=None, **opts):
    """(single store, multi store) with the same synthetic rows"""
    one, many = VecStore(dim), VecStore(dim, devices=devs)
    for s in (one, many):
        if chunk_size:
            s.set_chunk_size(chunk_size)
        for k, v in opts.items():
            s.set_option(k, v)
        if reserve:
            s.reserve(n)
        s.append_random(n, seed)
    return one, many


@pytest.mark.parametrize("devs", DEVS, ids=lambda d: f"x{len(d)}")
def test_multi_equals_single_store(oracle, devs):
    n, dim = 50_000, 96
    one, many = pair(dim, devs, n, seed=21)
    sh = many.shards()
    assert len(sh) == len(devs) and sum(c for _, _, c in sh) == n and all(f % 8 == 0 for _, f, _ in sh)
    assert max(c for _, _, c in sh) <= n // len(devs) + 1024  # the reserve planned an even split
    assert many.len() == n and np.array_equal(many.rows(1000, 3000), one.rows(1000, 3000))
    assert np.array_equal(many.inv_norms().view(np.uint32), one.inv_norms().view(np.uint32))
    rng = np.random.default_rng(5)
    q1 = rng.uniform(-1, 1, dim).astype(np.float32)
    q5 = rng.uniform(-1, 1, (5, dim)).astype(np.float32)
    rows = oracle.rand_rows(0, n, dim, 21)
    for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
        for k in (1, 10, 64, 100, 130, 300, 512, 513, 2000):
            for q in (q1, q5):
                a, ca = one.query(q, metric).take(k).collect_arrays()
                b, cb = many.query(q, metric).take(k).collect_arrays()
                same_hits(b, a, (metric, k, q.shape))
                assert ca == cb and b.size == k
        # against the oracle outright
        got, _ = many.query(q5, metric).take(40).collect_arrays()
        ref = oracle.vec_query(rows, q5, int(metric), 0 if metric == Metric.Euclidean else 1, 40, ties=oracle.TIES_CANONICAL)
        same_hits(got, ref, ("oracle", metric))
        # PER_QUERY
        for k in (7, 100, 600):
            a, ca = one.query(q5, metric).per_query().take(k).collect_arrays()
            b, cb = many.query(q5, metric).per_query().take(k).collect_arrays()
            same_hits(b, a, ("perq", metric, k))
            assert ca == cb == [k] * 5
    st = many.last_stats
    assert st["vectors_compared"] == 5 * n and st["total_chunks"] == (n + 1023) // 1024
    assert many.transport() == expected_transport()
    # score filters, take_min / take_max, empty results
    for thr, cmp in ((0.05, Cmp.Gt), (-0.02, Cmp.Lte), (0.9, Cmp.Gt)):
        a, _ = one.query(q5, Metric.Cosine).filter(thr, cmp).take_min(50).collect_arrays()
        b, _ = many.query(q5, Metric.Cosine).filter(thr, cmp).take_min(50).collect_arrays()
        same_hits(b, a, ("filter", thr, cmp))
    # host row mask (shorter than the store: rows beyond it are kept, src/vec.rs:234) — sliced per shard
    mask = rng.random(n - 4321) < 0.3
    for k in (10, 700):
        a, _ = one.query(q5, Metric.DotProduct).with_row_mask(mask).take(k).collect_arrays()
        b, _ = many.query(q5, Metric.DotProduct).with_row_mask(mask).take(k).collect_arrays()
        same_hits(b, a, ("row mask", k))
    # the matrix-core cascade on every shard, forced and by AUTO
    q40 = rng.uniform(-1, 1, (40, dim)).astype(np.float32)
    for path in (Path.Mfma, Path.Auto):
        a, _ = one.query(q40, Metric.Cosine).take(50).with_path(Path.Exact).collect_arrays()
        b, _ = many.query(q40, Metric.Cosine).take(50).with_path(path).collect_arrays()
        same_hits(b, a, ("cascade", path))
        b, cb = many.query(q40, Metric.Euclidean).per_query().take(20).with_path(path).collect_arrays()
        a, ca = one.query(q40, Metric.Euclidean).per_query().take(20).with_path(Path.Exact).collect_arrays()
        same_hits(b, a, ("cascade perq", path))
    assert many.last_stats["path_used"] == 2
    # the default take (src/vec.rs:213: every row), merged and per query
    small_one, small_many = pair(24, devs, 3000, seed=8, reserve=False)
    qs = rng.uniform(-1, 1, (3, 24)).astype(np.float32)
    a, _ = small_one.query(qs[0], Metric.DotProduct).collect_arrays()
    b, _ = small_many.query(qs[0], Metric.DotProduct).collect_arrays()
    same_hits(b, a, "default take")
    assert b.size == 3000
    a, _ = small_one.query(qs, Metric.Cosine).collect_arrays()
    b, _ = small_many.query(qs, Metric.Cosine).collect_arrays()
    same_hits(b, a, "default take, batch")
    for s in (one, many, small_one, small_many):
        s.close()


def test_multi_appends_without_a_plan_are_rebalanced(oracle):
    """No reserve: the appends land in the first shard; before the first query the rows are moved between the shards (and
    again when later appends unbalance them); appends of host rows cross shard boundaries; write_rows routes by row range."""
    dim, devs = 33, [0, 0, 0, 0]
    rng = np.random.default_rng(17)
    rows = rng.uniform(-1, 1, (20_000, dim)).astype(np.float32)
    one, many = VecStore(dim), VecStore(dim, devices=devs)
    one.add_vectors(rows[:9000])
    many.add_vectors(rows[:9000])
    assert [c for _, _, c in many.shards()] == [9000, 0, 0, 0]
    q = rng.uniform(-1, 1, (3, dim)).astype(np.float32)
    a, _ = one.query(q, Metric.Cosine).take(25).collect_arrays()
    b, _ = many.query(q, Metric.Cosine).take(25).collect_arrays()
    same_hits(b, a, "after the first rebalance")
    cnt = [c for _, _, c in many.shards()]
    assert sum(cnt) == 9000 and max(cnt) <= 9000 // 4 + 1024 and min(cnt) > 0, cnt
    assert np.array_equal(many.rows(), rows[:9000])
    # more rows: they go to the last shard, and the next query evens the shards out again
    for lo, hi in ((9000, 9001), (9001, 12_000), (12_000, 20_000)):
        one.add_vectors(rows[lo:hi])
        many.add_vectors(rows[lo:hi])
    a, _ = one.query(q, Metric.Euclidean).take(300).collect_arrays()
    b, _ = many.query(q, Metric.Euclidean).take(300).collect_arrays()
    same_hits(b, a, "after the second rebalance")
    cnt = [c for _, _, c in many.shards()]
    assert sum(cnt) == 20_000 and max(cnt) <= 5000 + 1024, cnt
    assert np.array_equal(many.rows(), rows) and np.array_equal(many.inv_norms().view(np.uint32), oracle.inv_norms(rows).view(np.uint32))
    # write_rows across a shard boundary
    first = many.shards()[2][1] - 3
    new = rng.uniform(-1, 1, (7, dim)).astype(np.float32)
    one.write_rows(first, new)
    many.write_rows(first, new)
    a, _ = one.query(new[3], Metric.Cosine).take(5).collect_arrays()
    b, _ = many.query(new[3], Metric.Cosine).take(5).collect_arrays()
    same_hits(b, a, "write_rows")
    assert int(b["index"][0]) == first + 3
    # a chunk size that does not divide the boundaries: the rows move once more (chunks must not straddle GPUs)
    for s in (one, many):
        s.set_chunk_size(1000)
    assert all(f % 1000 == 0 for _, f, c in many.shards() if c)
    cm = np.zeros(20, bool)
    cm[[1, 4, 5, 11, 19]] = True
    rq = many.query(q, Metric.DotProduct).take(40).resolve()
    a, _, _ = one._run(rq, chunk_mask=cm)
    b, _, st = many._run(rq, chunk_mask=cm)
    same_hits(b, a, "chunk mask")
    assert st["evaluated_chunks"] == 5 and st["pruned_chunks"] == 15 and st["vectors_compared"] == 3 * 5000
    # rebalancing switched off: the rows stay where the appends put them, the result does not change
    lazy = VecStore(dim, devices=devs)
    lazy.set_option("multi_rebalance", 0)
    lazy.add_vectors(rows)
    b, _ = lazy.query(q, Metric.Euclidean).take(300).collect_arrays()
    a, _ = one.query(q, Metric.Euclidean).take(300).collect_arrays()
    same_hits(b, a, "no rebalance")
    assert [c for _, _, c in lazy.shards()] == [20_000, 0, 0, 0]
    for s in (one, many, lazy):
        s.close()


@pytest.mark.parametrize("devs", [[0, 0, 0], [0] * 8], ids=lambda d: f"x{len(d)}")
def test_multi_tiny_and_ragged_stores(oracle, devs):
    """fewer rows than shards, one row, chunk sizes that are not multiples of 8, empty shards in front and behind"""
    rng = np.random.default_rng(3)
    for n, dim, cs in ((1, 4, 1024), (5, 3, 2), (17, 8, 3), (100, 5, 7), (1000, 16, 100), (4099, 12, 24)):
        rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        one, many = VecStore(dim), VecStore(dim, devices=devs)
        for s in (one, many):
            s.set_chunk_size(cs)
            s.add_vectors(rows)
        q = rng.uniform(-1, 1, (2, dim)).astype(np.float32)
        for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
            for k in (1, 3, n, 2 * n + 5):
                a, _ = one.query(q, metric).take(k).collect_arrays()
                b, _ = many.query(q, metric).take(k).collect_arrays()
                same_hits(b, a, (n, dim, cs, metric, k))
            ref = oracle.vec_query(rows, q, int(metric), 0 if metric == Metric.Euclidean else 1, 5, ties=oracle.TIES_CANONICAL)
            got, _ = many.query(q, metric).take(5).collect_arrays()
            same_hits(got, ref, ("oracle", n, metric))
        n_chunks = (n + cs - 1) // cs
        cm = rng.random(n_chunks) < 0.5
        rq = many.query(q, Metric.DotProduct).take(4).resolve()
        a, _, _ = one._run(rq, chunk_mask=cm)
        b, _, _ = many._run(rq, chunk_mask=cm)
        same_hits(b, a, ("chunk mask", n, cs))
        one.close()
        many.close()
    empty = VecStore(6, devices=devs)
    assert empty.query(np.ones(6, np.float32), Metric.Cosine).take(3).collect() == []  # src/vec.rs:222, 270
    with pytest.raises(OttersError) as ei:
        empty.query(np.ones(5, np.float32), Metric.Cosine).take(3).collect()
    assert "Query vector length 5 does not match expected dimension 6" in str(ei.value)


@pytest.mark.parametrize("case", [c for c in load("vec_store_cases.json") if "metric" in c], ids=lambda c: c["name"])
def test_golden_vec_cases_on_a_multi_store(case):
    store = VecStore(case["dim"], devices=[0, 0, 0])
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


@pytest.mark.parametrize("case", [c for c in load("meta_cases.json") if "metric" in c], ids=lambda c: c["name"])
def test_golden_meta_cases_on_a_multi_store(oracle, case):
    """the reference's MetaStore tests with the store spread over three shards: zone statistics and row predicates run per
    shard on its slice of the columns"""
    from otters_amd import Column as Cn
    cols = [Cn(c["name"], DataType[c["dtype"]]).from_(c["values"]) for c in case["columns"]]
    meta = MetaStore.from_columns(cols, devices=[0, 0, 0]).with_vectors(case["vectors"]).with_chunk_size(case["chunk_size"]).build()
    single = build_meta_case(case, host_only=False)
    res = meta_plan_from_case(case, meta).collect()
    want = meta_plan_from_case(case, single).collect()
    exp = case["expect"]
    check_expect(res.indices, res.scores, {k: v for k, v in exp.items() if k != "stats"})
    assert res.indices == want.indices
    assert np.array_equal(np.array(res.scores, np.float32).view(np.uint32), np.array(want.scores, np.float32).view(np.uint32))
    a, b = meta.last_query_stats(), single.last_query_stats()
    assert (a.total_chunks, a.pruned_chunks, a.evaluated_chunks, a.vectors_compared) == (b.total_chunks, b.pruned_chunks, b.evaluated_chunks, b.vectors_compared)


def _meta_pair(n, dim, cs, seed, devs):
    rng = np.random.default_rng(seed)
    vec = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    chunk = np.arange(n) // cs

    def cols():
        r = np.random.default_rng(seed + 1)
        return [Column.from_numpy("price", DataType.Float64, (chunk % 5) * 20.0 + r.uniform(0, 25, n), r.random(n) < 0.05),
                Column.from_numpy("version", DataType.Int32, (chunk % 3) + r.integers(0, 2, n), r.random(n) < 0.05),
                Column.from_numpy("ts", DataType.DateTime, 1_700_000_000_000 + chunk.astype(np.int64) * 86_400_000 + r.integers(0, 86_400_000, n)),
                Column.from_numpy("w", DataType.Float32, r.normal(0, 1, n).astype(np.float32), r.random(n) < 0.02),
                Column.from_numpy("grade", DataType.String, np.array(["A", "B", "C", "D"])[(chunk + r.integers(0, 2, n)) % 4], r.random(n) < 0.03)]
    one = MetaStore.from_columns(cols()).with_vectors(vec).with_chunk_size(cs).build()
    many = MetaStore.from_columns(cols(), devices=devs).with_vectors(vec).with_chunk_size(cs).build()
    return one, many, vec


@pytest.mark.parametrize("devs", DEVS, ids=lambda d: f"x{len(d)}")
def test_multi_metastore_filters_masks_and_zone_stats(devs):
    n, dim, cs = 30_011, 48, 256
    one, many, vec = _meta_pair(n, dim, cs, 31, devs)
    # zone statistics built per shard on the GPU equal the single store's
    for name in ("price", "version", "ts", "w"):
        za, zb = one._zones[name], many._zones[name]
        assert np.array_equal(za.min, zb.min) and np.array_equal(za.max, zb.max) and np.array_equal(za.non_null, zb.non_null), name
    filters = [
        lambda: col("price").lt(50.0) & col("version").gte(2),
        lambda: (col("price").lte(30.0) | col("price").gt(90.0)) & col("w").gt(-0.5),
        lambda: col("version").neq(1) & col("ts").gte("2023-11-20"),
        lambda: col("grade").eq("A") | col("grade").eq("B"),
    ]
    rng = np.random.default_rng(2)
    q = rng.uniform(-1, 1, (4, dim)).astype(np.float32)
    for f in filters:
        compiled = f().compile(many.schema())
        if many._device_mask_ok(compiled):  # the device row mask, evaluated per shard and stitched together
            ma = one.build_row_mask_device(compiled, fetch=True)
            mb = many.build_row_mask_device(compiled, fetch=True)
            assert np.array_equal(ma, mb) and np.array_equal(mb, many.build_row_mask_host(compiled))
        for metric in (Metric.Cosine, Metric.Euclidean):
            for k in (10, 200, 900):
                for plan_of in (lambda m: m.query(q[0], metric), lambda m: m.query_batch(q, metric)):
                    ra = plan_of(one).meta_filter(f()).vec_filter(-0.2, Cmp.Gt).take(k).collect()
                    rb = plan_of(many).meta_filter(f()).vec_filter(-0.2, Cmp.Gt).take(k).collect()
                    assert ra.indices == rb.indices and len(rb.indices) > 0
                    assert np.array_equal(np.array(ra.scores, np.float32).view(np.uint32), np.array(rb.scores, np.float32).view(np.uint32))
                    sa, sb = one.last_query_stats(), many.last_query_stats()
                    assert (sa.total_chunks, sa.pruned_chunks, sa.evaluated_chunks, sa.vectors_compared) == \
                           (sb.total_chunks, sb.pruned_chunks, sb.evaluated_chunks, sb.vectors_compared)
    # the default take of a MetaStore query (src/meta.rs:638-644) with a filter
    ra = one.query(q[1], Metric.DotProduct).meta_filter(filters[0]()).collect()
    rb = many.query(q[1], Metric.DotProduct).meta_filter(filters[0]()).collect()
    assert ra.indices == rb.indices and len(rb.indices) > 1000


@pytest.mark.parametrize("devs", DEVS, ids=lambda d: f"x{len(d)}")
def test_multi_reference_tie_orders(oracle, devs):
    """tie_order 1 (ONE collector over the whole store) and 2 (per-chunk collectors, concat-sort-truncate) across shards:
    quantised rows, so nearly every cut runs through a group of equal scores that spans several shards.  Checked against the
    oracle's literal restatement of the collector."""
    rng = np.random.default_rng(404)
    for n, dim, nq, cs in ((4000, 6, 3, 64), (20_011, 8, 2, 256), (203, 3, 1, 8)):
        rows = rng.integers(-2, 3, (n, dim)).astype(np.float32)
        queries = rng.integers(-2, 3, (nq, dim)).astype(np.float32)
        queries[np.all(queries == 0, axis=1)] = 1.0
        many = VecStore(dim, devices=devs)
        many.set_chunk_size(cs)
        many.set_tie_order("reference")
        many.add_vectors(rows)
        one = VecStore(dim)
        one.set_chunk_size(cs)
        one.set_tie_order("reference")
        one.add_vectors(rows)
        mask = rng.random(n) < 0.85
        for metric, take in ((Metric.DotProduct, 1), (Metric.Euclidean, 0), (Metric.Cosine, 1)):
            for k in (1, 7, 20, 64, 150, 511, 512, 700):
                if k > n * nq:
                    continue
                for m in (None, mask):
                    pa, pb = one.query(queries, metric), many.query(queries, metric)
                    if m is not None:
                        pa, pb = pa.with_row_mask(m), pb.with_row_mask(m)
                    a, _ = (pa.take_max(k) if take else pa.take_min(k)).collect_arrays()
                    b, _ = (pb.take_max(k) if take else pb.take_min(k)).collect_arrays()
                    same_hits(b, a, ("tie_order 1", n, metric, k, m is not None))
                    lit = oracle.vec_query(rows, queries, int(metric), take, k, row_mask=m, ties=oracle.TIES_LITERAL)
                    assert np.array_equal(b["score"].view(np.uint32), lit["score"].view(np.uint32))
                    assert sorted(zip(b["index"].tolist(), b["query"].tolist())) == sorted(zip(lit["index"].tolist(), lit["query"].tolist()))
            for k in (5, 40):
                a, ca = one.query(queries, metric).per_query().take(k).collect_arrays()
                b, cb = many.query(queries, metric).per_query().take(k).collect_arrays()
                same_hits(b, a, ("tie_order 1 perq", n, metric, k))
        # MetaStore semantics: one collector per chunk
        for s in (one, many):
            s.set_tie_order("reference_chunked")
        n_chunks = (n + cs - 1) // cs
        cm = (np.arange(n_chunks) % 3) != 1
        for metric, take in ((Metric.DotProduct, 1), (Metric.Euclidean, 0), (Metric.Cosine, 1)):
            for k in (1, 4, 10, 30, 100):
                for chunk_mask in (None, cm):
                    rq = (many.query(queries, metric).take_max(k) if take else many.query(queries, metric).take_min(k)).resolve()
                    a, _, _ = one._run(rq, chunk_mask=chunk_mask)
                    b, _, _ = many._run(rq, chunk_mask=chunk_mask)
                    same_hits(b, a, ("tie_order 2", n, metric, k, chunk_mask is not None))
                    lit, _ = oracle.meta_query(rows, cs, queries, int(metric), take, k, chunk_mask=chunk_mask, ties=oracle.TIES_LITERAL)
                    assert np.array_equal(b["score"].view(np.uint32), lit["score"].view(np.uint32))
                    assert sorted(b["index"].tolist()) == sorted(lit["index"].tolist())
        one.close()
        many.close()


@pytest.mark.parametrize("devs", [[0, 0, 0], [0] * 8], ids=lambda d: f"x{len(d)}")
def test_multi_large_results_are_merged_by_all_shard_threads(devs):
    """k > 512: every shard's sorted list comes to the host and the shard threads merge them together — few large groups are cut
    into equal ranges by rank (each thread finds its own cuts by binary search over the lists), many small groups are dealt
    out whole; results below 32768 hits stay on the calling thread.  All three against one store, both tie orders of the key."""
    n, dim = 70_001, 40
    one, many = pair(dim, devs, n, seed=77)
    rng = np.random.default_rng(9)
    q1 = rng.uniform(-1, 1, dim).astype(np.float32)
    q3 = rng.uniform(-1, 1, (3, dim)).astype(np.float32)
    q20 = rng.uniform(-1, 1, (20, dim)).astype(np.float32)
    for tie in ("canonical", "reference"):
        for s in (one, many):
            s.set_tie_order(tie)
        for metric in (Metric.Cosine, Metric.Euclidean):
            plans = [
                ("default take, one query: one group cut into ranges", lambda s: s.query(q1, metric)),
                ("default take, 3 queries merged: 210k hits, one group", lambda s: s.query(q3, metric)),
                ("take 40000 of 3 queries merged", lambda s: s.query(q3, metric).take(40_000)),
                ("per query, 3 groups of 70k: fewer groups than shards or large groups", lambda s: s.query(q3, metric).per_query()),
                ("per query, 20 groups of 3000: whole groups per thread", lambda s: s.query(q20, metric).per_query().take(3000)),
                ("per query, 20 groups of 600: below the threshold", lambda s: s.query(q20, metric).per_query().take(600)),
                ("filter leaves the shards uneven lists", lambda s: s.query(q3, metric).filter(0.0, Cmp.Gt).per_query()),
            ]
            for name, plan in plans:
                a, ca = plan(one).collect_arrays()
                b, cb = plan(many).collect_arrays()
                same_hits(b, a, (name, tie, metric))
                assert ca == cb
    # a row mask that empties some shards entirely
    mask = np.zeros(n, bool)
    mask[: n // 3] = True
    a, _ = one.query(q3, Metric.DotProduct).with_row_mask(mask).collect_arrays()
    b, _ = many.query(q3, Metric.DotProduct).with_row_mask(mask).collect_arrays()
    same_hits(b, a, "row mask, default take")
    assert b.size == 3 * (n // 3)
    one.close()
    many.close()


def test_multi_small_stores_stay_on_one_shard(oracle):
    """option multi_min_shard_rows (default 32768; the suite runs with 0): a store of fewer than twice that many rows stays on its
    first GPU and is answered by that shard alone — no fan-out, no exchange, the transport is never even chosen; as the store
    grows a shard is brought in per 32768 rows, and a plan (ott_store_reserve) lays out for the planned size."""
    dim, devs = 24, [0] * 8
    rng = np.random.default_rng(5)
    rows = rng.uniform(-1, 1, (120_000, dim)).astype(np.float32)
    one, many = VecStore(dim), VecStore(dim, devices=devs)
    many.set_option("multi_min_shard_rows", 32768)
    q = rng.uniform(-1, 1, (3, dim)).astype(np.float32)

    def compare(where):
        for s in (one, many):
            s.set_tie_order("canonical")
        for k in (10, 700):
            a, ca = one.query(q, Metric.Cosine).take(k).collect_arrays()
            b, cb = many.query(q, Metric.Cosine).take(k).collect_arrays()
            same_hits(b, a, (where, k))
            a, ca = one.query(q, Metric.Euclidean).per_query().take(k).collect_arrays()
            b, cb = many.query(q, Metric.Euclidean).per_query().take(k).collect_arrays()
            same_hits(b, a, (where, "per query", k))
            assert ca == cb
        mask = rng.random(one.len() - 17) < 0.4
        a, _ = one.query(q, Metric.DotProduct).with_row_mask(mask).take(50).collect_arrays()
        b, _ = many.query(q, Metric.DotProduct).with_row_mask(mask).take(50).collect_arrays()
        same_hits(b, a, (where, "row mask"))
        for s in (one, many):
            s.set_tie_order("reference")
        a, _ = one.query(q, Metric.Cosine).take(33).collect_arrays()
        b, _ = many.query(q, Metric.Cosine).take(33).collect_arrays()
        same_hits(b, a, (where, "reference ties"))

    for s in (one, many):
        s.add_vectors(rows[:10_000])
    compare("10k rows")
    assert [c for _, _, c in many.shards()] == [10_000] + [0] * 7
    assert many.transport() == "undecided"  # nothing was ever exchanged
    assert many.last_stats["vectors_compared"] == 3 * 10_000
    for s in (one, many):
        s.add_vectors(rows[10_000:70_000])
    compare("70k rows")
    cnt = [c for _, _, c in many.shards()]
    assert sum(cnt) == 70_000 and cnt[2:] == [0] * 6 and min(cnt[:2]) > 30_000, cnt
    assert many.transport() == expected_transport()
    for s in (one, many):
        s.add_vectors(rows[70_000:])
    compare("120k rows")
    cnt = [c for _, _, c in many.shards()]
    assert sum(cnt) == 120_000 and cnt[3:] == [0] * 5 and min(cnt[:3]) > 30_000, cnt
    assert np.array_equal(many.rows(), rows)
    # a plan lays out for the planned size, whatever is there yet
    planned = VecStore(dim, devices=devs)
    planned.set_option("multi_min_shard_rows", 32768)
    planned.reserve(400_000)
    planned.add_vectors(rows[:60_000])
    cnt = [c for _, _, c in planned.shards()]
    assert cnt[0] == 400_000 // 8 // 1024 * 1024 or abs(cnt[0] - 50_000) <= 1024, cnt
    assert cnt[1] == 60_000 - cnt[0] and cnt[2:] == [0] * 6, cnt
    small_plan = VecStore(dim, devices=devs)
    small_plan.set_option("multi_min_shard_rows", 32768)
    small_plan.reserve(20_000)
    small_plan.add_vectors(rows[:20_000])
    assert [c for _, _, c in small_plan.shards()] == [20_000] + [0] * 7
    b, _ = small_plan.query(q, Metric.Cosine).take(5).collect_arrays()
    ref = oracle.vec_query(rows[:20_000], q, int(Metric.Cosine), 1, 5, ties=oracle.TIES_CANONICAL)
    same_hits(b, ref, "small plan against the oracle")
    for s in (one, many, planned, small_plan):
        s.close()


def test_multi_c4_shape_cascade():
    """config 4's shape in small on ONE process: 1024 queries, cosine top-100 per query, every shard runs the matrix-core
    cascade, the exchange carries [1024][128] slots per shard, one grouped device merge"""
    n, dim = 60_000, 64
    one, many = pair(dim, [0] * 8, n, seed=11)
    big = np.random.default_rng(12).uniform(-1, 1, (1024, dim)).astype(np.float32)
    b, cb = many.query(big, Metric.Cosine).per_query().take(100).with_path(Path.Mfma).collect_arrays()
    assert many.last_stats["path_used"] == 2 and cb == [100] * 1024
    a, ca = one.query(big, Metric.Cosine).per_query().take(100).with_path(Path.Exact).collect_arrays()
    same_hits(b, a, "c4 shape")
    b, _ = many.query(big[:256], Metric.Cosine).take(100).collect_arrays()
    a, _ = one.query(big[:256], Metric.Cosine).take(100).with_path(Path.Exact).collect_arrays()
    same_hits(b, a, "c2 shape, merged")
    one.close()
    many.close()


def test_multi_rccl_transport_on_one_device():
    """the RCCL exchange of the in-process store for real — ncclCommInitAll, ncclGroupStart / ncclAllGather / ncclGroupEnd on
    the shard's stream, the merge behind it — on the device list RCCL accepts on this box: [0] (a second rank on the same GPU
    is refused by ncclCommInitAll itself, which the test also shows)"""
    if multi_mode() != "local":
        pytest.skip("the real RCCL on the device list [0]: once, in the mode without stand-ins")
    N.lib()
    N.preload_torch_rccl()
    n, dim = 20_000, 40
    one = VecStore(dim)
    one.append_random(n, 4)
    many = VecStore(dim, devices=[0])
    many.set_option("multi_transport", 2)
    many.append_random(n, 4)
    q = np.random.default_rng(1).uniform(-1, 1, (6, dim)).astype(np.float32)
    for k in (10, 300):
        a, _ = one.query(q, Metric.Cosine).take(k).collect_arrays()
        b, _ = many.query(q, Metric.Cosine).take(k).collect_arrays()
        same_hits(b, a, ("rccl", k))
        a, _ = one.query(q, Metric.Euclidean).per_query().take(k).collect_arrays()
        b, _ = many.query(q, Metric.Euclidean).per_query().take(k).collect_arrays()
        same_hits(b, a, ("rccl perq", k))
    assert many.transport() == "rccl"
    dup = VecStore(dim, devices=[0, 0])
    dup.set_option("multi_transport", 2)
    dup.append_random(100, 1)
    with pytest.raises(OttersError) as ei:
        dup.query(q[0], Metric.Cosine).take(3).collect()
    assert "distinct device ordinals" in str(ei.value)
    for s in (one, many, dup):
        s.close()


def test_multi_concurrent_queries_from_threads():
    """the reference's query is `&self` (src/vec.rs:387): several host threads query one multi-GPU store at once"""
    import threading
    n, dim = 40_000, 64
    one, many = pair(dim, [0, 0, 0, 0], n, seed=2)
    rng = np.random.default_rng(8)
    qs = rng.uniform(-1, 1, (12, dim)).astype(np.float32)
    want = [one.query(qs[i], Metric.Cosine).take(20).collect_arrays()[0] for i in range(12)]
    errs = []

    def worker(i):
        try:
            for _ in range(15):
                got, _ = many.query(qs[i], Metric.Cosine).take(20).collect_arrays()
                same_hits(got, want[i], i)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(12)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    one.close()
    many.close()


@pytest.mark.parametrize("devs", [None, [0, 0, 0]], ids=["one_gpu", "three_shards"])
def test_queries_from_threads_while_rows_are_appended(oracle, devs):
    """Readers and a writer at once: four threads query in a loop (single queries and small batches) while the main thread keeps
    appending — single rows (staged), small and large pieces — so that staged flushes, row moves between shards, the background
    plane builder and the query contexts all meet.  Every answer a reader gets must be the exact top-k of SOME prefix of the
    rows (appends are atomic: a query sees all rows of an append or none), and the final store equals one built in one go."""
    import threading
    dim, n_total = 48, 60_000
    rows = oracle.rand_rows(0, n_total, dim, 77)
    store = VecStore(dim, devices=devs)
    store.set_option("hi_prebuild", 1)  # the builder runs after every append, whatever the size
    store.add_vectors(rows[:2000])
    qs = oracle.rand_rows(0, 6, dim, 78)
    stop = threading.Event()
    errs, seen = [], [0]

    def reader(i):
        try:
            rng = np.random.default_rng(i)
            while not stop.is_set():
                q = qs[i % 6] if rng.random() < 0.6 else qs[rng.integers(0, 6, 3)]
                hits, _ = store.query(q, Metric.Cosine).take(8).collect_arrays()
                assert hits.size == 8 and np.all(np.diff(hits["score"]) <= 0)
                assert int(hits["index"].max()) < n_total
                seen[0] += 1
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=reader, args=(i,)) for i in range(4)]
    [t.start() for t in ths]
    at = 2000
    rng = np.random.default_rng(0)
    while at < n_total:
        step = int(rng.choice([1, 1, 3, 200, 5000, 9000]))
        step = min(step, n_total - at)
        if step == 1:
            store.add_vector(rows[at])
        else:
            store.add_vectors(rows[at:at + step])
        at += step
    stop.set()
    [t.join() for t in ths]
    assert not errs, errs[:2]
    # (the readers get in between the appends: since round 5 a waiting writer goes first — ott::host::RwGate — so they see fewer
    #  queries through during the load than under glibc's reader-preferring rwlock, where the APPENDS waited instead)
    assert seen[0] >= 4
    assert store.len() == n_total and np.array_equal(store.rows(), rows)
    for q in qs:
        got, _ = store.query(q, Metric.Cosine).take(50).collect_arrays()
        ref = oracle.vec_query(rows, q, oracle.METRIC_COSINE, oracle.TAKE_MAX, 50, ties=oracle.TIES_CANONICAL)
        same_hits(got, ref, "after concurrent load")
    store.close()


def test_multi_errors_leave_the_store_usable():
    """an allocation one shard cannot satisfy is an error (OTT_ERR_OOM, naming the shard), not a crash; the rows stay and the
    store keeps answering; misuse is reported like on a single-GPU store; rows cannot move once columns are resident"""
    dim = 768
    many = VecStore(dim, devices=[0, 0, 0, 0])
    many.append_random(4000, 3)
    q = np.random.default_rng(2).uniform(-1, 1, dim).astype(np.float32)
    want, _ = many.query(q, Metric.Cosine).take(5).collect_arrays()
    with pytest.raises(OttersError) as ei:
        many.reserve(800_000_000)  # 4 x 200M x 768 f32 = 2.4 TB
    assert getattr(ei.value, "status", None) == -3 and "out of memory" in str(ei.value), str(ei.value)  # OTT_ERR_OOM
    assert many.len() == 4000
    got, _ = many.query(q, Metric.Cosine).take(5).collect_arrays()
    same_hits(got, want, "after a failed reserve")
    many.append_random(100, 3)  # and it still grows
    assert many.len() == 4100 and sum(c for _, _, c in many.shards()) == 4100
    # capacity / argument errors through the C ABI
    buf = np.zeros(4, dtype=N.HIT_DTYPE)
    d = N.QueryDesc()
    d.queries, d.nq, d.metric, d.take, d.k = q.ctypes.data, 1, 0, 1, 10
    n_out = C.c_uint64(0)
    assert N.lib().ott_query(many._handle(), C.byref(d), N.ptr(buf), 4, C.byref(n_out), None, None) != 0 and b"capacity" in N.lib().ott_last_error()
    d.metric = 9
    assert N.lib().ott_query(many._handle(), C.byref(d), N.ptr(buf), 4, C.byref(n_out), None, None) != 0
    assert N.lib().ott_store_create_multi(8, 0, None, C.byref(C.c_void_p())) != 0
    ids = (C.c_int * 2)(0, 99)
    assert N.lib().ott_store_create_multi(8, 2, ids, C.byref(C.c_void_p())) != 0  # no such device
    # columns pin the rows: a chunk size that would need the rows moved is refused, and says why
    vals = np.arange(4100, dtype=np.int32)
    cid = C.c_uint32(0)
    N.check(N.lib().ott_store_add_column(many._handle(), 0, N.ptr(vals), None, 4100, C.byref(cid)))
    with pytest.raises(OttersError) as ei:
        many.set_chunk_size(1000)
    assert "metadata columns" in str(ei.value)
    got, _ = many.query(q, Metric.Cosine).take(5).collect_arrays()
    same_hits(got, want, "after a refused chunk size")
    many.close()


def test_multi_unsupported_calls_say_so():
    many = VecStore(8, devices=[0, 0])
    many.append_random(100, 1)
    buf = np.zeros(16, dtype=N.HIT_DTYPE)
    d = N.QueryDesc()
    q = np.ones(8, np.float32)
    d.queries, d.nq, d.k = q.ctypes.data, 1, 4
    rc = N.lib().ott_query_device(many._handle(), C.byref(d), N.ptr(buf), 16, None, None)
    assert rc == -4 and b"multi-GPU store" in N.lib().ott_last_error()
    many.close()


def test_the_exchange_took_the_branch_the_mode_names():
    """local: blocks land in the merging GPU's buffer directly; remote: every shard but the first goes through its send buffer and
    a peer copy; fake_rccl: the grouped all-gather — counted by the stand-in library itself"""
    mode = multi_mode()
    one, many = pair(32, [0, 0, 0], 6000, seed=3)
    q = np.random.default_rng(0).uniform(-1, 1, 32).astype(np.float32)
    before = C.CDLL(FAKE_RCCL).fake_rccl_gathers() if mode == "fake_rccl" else 0
    for _ in range(5):
        a, _ = one.query(q, Metric.Cosine).take(10).collect_arrays()
        b, _ = many.query(q, Metric.Cosine).take(10).collect_arrays()
        same_hits(b, a, mode)
    assert many.transport() == expected_transport()
    if mode == "fake_rccl":
        fn = C.CDLL(FAKE_RCCL).fake_rccl_gathers
        fn.restype = C.c_uint64
        assert fn() - before == 5, (fn(), before)
    one.close()
    many.close()
