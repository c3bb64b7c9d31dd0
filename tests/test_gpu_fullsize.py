"""GPU parity at BASELINE.json's full sizes.  Config 1 (1M x 128) is small enough for a direct
oracle comparison; at 10M x 768 (configs 2/3 and the headline metric) the oracle cannot score the
corpus in seconds, so parity goes through size-independent properties: planted near-duplicates
must come back, every returned score is re-derived by the oracle from regenerated rows, a
sampled completeness check, top-k(all) == merge(top-k(even chunks), top-k(odd chunks)), the
two independent GPU paths (exact-order VALU vs MFMA + re-score) must agree bit for bit, and a row-level
predicate's top-k must be the satisfying prefix of the unfiltered ranking."""
import os

import numpy as np
import pytest

from otters_amd import Cmp, Column, DataType, MetaStore, Metric, Path, VecStore, col

pytestmark = pytest.mark.gpu

SEED = 0x07735


def test_config1_1Mx128_dot_top10_direct_oracle(oracle):
    n, dim = 1_000_000, 128
    store = VecStore(dim)
    store.append_random(n, SEED)
    rows = oracle.rand_rows(0, n, dim, SEED)
    q = oracle.rand_rows(0, 1, dim, SEED + 1)[0]
    res = store.query(q, Metric.DotProduct).take(10).collect()
    ref = oracle.vec_query(rows, q, oracle.METRIC_DOT, oracle.TAKE_MAX, 10, ties=oracle.TIES_CANONICAL, fast=True)
    assert [r.index for r in res] == [int(i) for i in ref["index"]]
    assert np.array_equal(np.array([r.score for r in res], np.float32).view(np.uint32), ref["score"].view(np.uint32))
    lit = oracle.vec_query(rows, q, oracle.METRIC_DOT, oracle.TAKE_MAX, 10, ties=oracle.TIES_LITERAL, fast=True)
    assert np.array_equal(lit["index"], ref["index"])  # no ties in random data: literal collector == canonical
    assert store.last_stats["bytes_scanned"] == n * dim * 4


@pytest.fixture(scope="module")
def big():
    n, dim, cs = 10_000_000, 768, 4096
    bucket = Column.from_numpy("bucket", DataType.Int32, ((np.arange(n) // cs) % 2).astype(np.int32))
    meta = MetaStore.from_columns([bucket]).with_random_vectors(n, dim, SEED).with_chunk_size(cs).build()
    return meta, n, dim, cs


def _oracle_scores(oracle, store, idx, q, dim):
    out = []
    for i in idx:
        row = oracle.rand_rows(int(i), 1, dim, SEED)[0]
        assert np.array_equal(store.rows(int(i), 1)[0], row)
        out.append(oracle.cosine(q, row, oracle.inv_norms(q)[0], oracle.inv_norms(row)[0]))
    return np.array(out, np.float32)


def test_headline_10Mx768_cosine_top10_properties(oracle, big):
    meta, n, dim, cs = big
    store = meta._store
    q = oracle.rand_rows(0, 1, dim, SEED + 1)[0]
    res = store.query(q, Metric.Cosine).take(10).collect()
    idx = np.array([r.index for r in res])
    sc = np.array([r.score for r in res], np.float32)
    assert len(res) == 10 and np.all(np.diff(sc) <= 0)
    # every returned score is the oracle's, bit for bit
    assert np.array_equal(sc.view(np.uint32), _oracle_scores(oracle, store, idx, q, dim).view(np.uint32))
    # sampled completeness: no row in a 200k-row sample beats the k-th score unless it is in the result
    rng = np.random.default_rng(0)
    for start in rng.integers(0, n - 50_000, 4):
        blk = oracle.rand_rows(int(start), 50_000, dim, SEED)
        s = oracle.vec_query(blk, q, oracle.METRIC_COSINE, oracle.TAKE_MAX, 1, fast=True)
        assert s["score"][0] <= sc[-1] or (int(s["index"][0]) + int(start)) in set(idx.tolist())
    # merge property over a chunk partition (the reference's per-chunk top-k then merge, src/meta.rs:693-709)
    n_chunks = (n + cs - 1) // cs
    even = (np.arange(n_chunks) % 2) == 0
    rq = store.query(q, Metric.Cosine).take(10).resolve()
    h_even, _, st_e = store._run(rq, chunk_mask=even)
    h_odd, _, st_o = store._run(rq, chunk_mask=~even)
    both = np.concatenate([h_even, h_odd])
    order = np.lexsort((both["index"], -both["score"].astype(np.float64)))
    assert np.array_equal(both[order][:10]["index"], idx)
    assert st_e["vectors_compared"] + st_o["vectors_compared"] == n
    assert st_e["bytes_scanned"] + st_o["bytes_scanned"] == n * (dim * 4 + 4)


def test_config3_meta_prune_vecfilter_planted(oracle, big):
    meta, n, dim, cs = big
    store = meta._store
    q = oracle.rand_rows(0, 1, dim, SEED + 1)[0]
    rng = np.random.default_rng(3)
    planted = np.arange(12_345, n, 156_007)[:64]
    dup = (q + rng.normal(0, 0.05, (planted.size, dim))).astype(np.float32)
    saved = [store.rows(int(i), 1) for i in planted]
    for i, r in zip(planted, dup):
        store.write_rows(int(i), r[None, :])
    try:
        res = meta.query(q, Metric.Cosine).meta_filter(col("bucket").eq(1)).vec_filter(0.5, Cmp.Gt).take(10).collect()
        st = meta.last_query_stats()
        n_chunks = (n + cs - 1) // cs
        assert st.total_chunks == n_chunks == 2442 and st.pruned_chunks == 1221 and st.evaluated_chunks == 1221
        assert st.vectors_compared == n - 1221 * cs  # rows of the surviving odd chunks (the short last chunk is odd)
        kept = planted[(planted // cs) % 2 == 1]
        want = sorted(((float(oracle.cosine(q, dup[list(planted).index(i)], oracle.inv_norms(q)[0], oracle.inv_norms(dup[list(planted).index(i)])[0])), int(i)) for i in kept), key=lambda t: (-t[0], t[1]))[:10]
        assert res.indices == [i for _, i in want]
        assert np.array_equal(np.array(res.scores, np.float32).view(np.uint32), np.array([s for s, _ in want], np.float32).view(np.uint32))
        assert all(s > 0.5 for s in res.scores)
        # random 768-d rows never reach 0.5: without the planted rows' chunks nothing qualifies
        res2 = meta.query(q, Metric.Cosine).meta_filter(col("bucket").eq(0)).vec_filter(0.5, Cmp.Gt).take(10).collect()
        assert set(res2.indices) == set(planted[(planted // cs) % 2 == 0][:10].tolist()) or len(res2) == min(10, (planted // cs % 2 == 0).sum())
    finally:
        for i, r in zip(planted, saved):
            store.write_rows(int(i), r)


def test_config2_batch_paths_agree_at_full_size(oracle, big):
    meta, n, dim, cs = big
    store = meta._store
    queries = oracle.rand_rows(0, 64, dim, SEED + 1)
    a = store.query(queries, Metric.Cosine).take(100).with_path(Path.Mfma).per_query().collect()
    assert store.last_stats["path_used"] == 2
    b = store.query(queries[:8], Metric.Cosine).take(100).with_path(Path.Exact).per_query().collect()
    for i in range(8):
        assert a[i] == b[i]
    # merged (reference semantics) == canonical merge of the per-query lists
    m = store.query(queries, Metric.Cosine).take(100).with_path(Path.Mfma).collect()
    flat = sorted(((r.score, r.index, qi) for qi, lst in enumerate(a) for r in lst), key=lambda t: (-t[0], t[1], t[2]))[:100]
    assert [(r.score, r.index) for r in m] == [(s, i) for s, i, _ in flat]
    # one of the returned rows re-derived by the oracle
    r0 = a[5][0]
    row = oracle.rand_rows(r0.index, 1, dim, SEED)[0]
    assert np.float32(r0.score) == oracle.cosine(queries[5], row, oracle.inv_norms(queries[5])[0], oracle.inv_norms(row)[0])


def test_row_level_predicate_at_full_size(oracle, big):
    """A row-level predicate with no zonemap help (v = row % 7 == 3: every chunk survives, one row in seven passes), evaluated
    on the GPU over 10M rows, on both scoring paths.  The filtered top-k must be exactly the first k rows of the UNFILTERED
    ranking that satisfy the predicate (the unfiltered top-2000 holds ~285 of them), every hit must satisfy it, and the
    exact and MFMA paths must agree bit for bit."""
    meta, n, dim, cs = big
    store = meta._store
    v = Column.from_numpy("v", DataType.Int32, (np.arange(n) % 7).astype(np.int32))
    m2 = MetaStore.from_columns([v]).with_random_vectors(n, dim, SEED).with_chunk_size(cs).build(_host_only=True)  # columns + zonemaps only ...
    m2._store = store                                                            # ... over the vectors already resident in HBM
    q = oracle.rand_rows(0, 1, dim, SEED + 2)[0]
    k = 25
    res = m2.query(q, Metric.Cosine).meta_filter(col("v").eq(3)).take(k).collect()
    st = m2.last_query_stats()
    assert st.pruned_chunks == 0 and st.evaluated_chunks == st.total_chunks
    assert len(res) == k and all(i % 7 == 3 for i in res.indices)
    top, _ = store.query(q, Metric.Cosine).take(2000).with_path(Path.Exact).collect_arrays()
    keep = top[top["index"] % 7 == 3][:k]
    assert len(keep) == k
    assert res.indices == [int(i) for i in keep["index"]]
    assert np.array_equal(np.array(res.scores, np.float32).view(np.uint32), keep["score"].view(np.uint32))
    # batch of queries through the MFMA path with the same device row mask == the exact path, query by query
    Q = oracle.rand_rows(0, 24, dim, SEED + 3)
    a = m2.query_batch(Q, Metric.Cosine).meta_filter(col("v").eq(3)).take(k).with_path(Path.Mfma).collect()
    assert store.last_stats["path_used"] == 2
    b = m2.query_batch(Q, Metric.Cosine).meta_filter(col("v").eq(3)).take(k).with_path(Path.Exact).collect()
    assert a.indices == b.indices
    assert np.array_equal(np.array(a.scores, np.float32).view(np.uint32), np.array(b.scores, np.float32).view(np.uint32))
    assert all(i % 7 == 3 for i in a.indices)


# (the hi plane is IEEE half by default since round 3; "hi_fmt": 0 rebuilds it as bf16 — which is also what the opt-in
# phase-staggered 256-query kernel reads)
C2_MODES = {"default_cascade": {}, "split_pass_only": {"no_hi_pass": 1}, "f32_pipe": {"mfma_f32": 1}, "bf16_plane": {"hi_fmt": 0},
            "int8_first": {"hi_fmt": 2}}  # round 5: an int8 plane (a quarter of the f32 bytes, exact i32 accumulation) as the cascade's first level


@pytest.mark.parametrize("mode", list(C2_MODES), ids=list(C2_MODES))
def test_config2_real_shape_256_queries_top100(oracle, big, mode):
    """BASELINE config 2 at its real shape — 10M x 768, 256 queries, cosine, take(100) — through every candidate pass of the
    batch path (default cascade: bf16 hi pass first; split-bf16 pass alone; f32 matrix pipe; the phase-staggered hi-pass
    kernel).  No query may need a later level or the exact path; 8 sampled queries equal the exact-order kernel bit for bit;
    every hit of 2 queries is re-derived by the oracle from the regenerated row; a sampled completeness check; the merged
    (reference-semantics) result is the canonical merge of the per-query lists; and the certification's error bound is used to
    less than half (ott_stats.err_ratio_max)."""
    meta, n, dim, cs = big
    store = meta._store
    nq, k = 256, 100
    queries = oracle.rand_rows(0, nq, dim, SEED + 1)
    opts = C2_MODES[mode]
    for name, v in opts.items():
        store.set_option(name, v)
    if "hi_fmt" in opts:  # the plane's format is fixed when it is built: drop it, the next batch builds it anew
        store.set_batch_image(False)
        store.set_batch_image(True)
    try:
        hits, counts = store.query(queries, Metric.Cosine).take(k).with_path(Path.Mfma).per_query().collect_arrays()
        st = dict(store.last_stats)
        assert st["path_used"] == 2 and counts == [k] * nq
        assert st["refined"] == 0 and st["retries"] == 0, st   # certified by the first pass, all 256 queries
        if mode == "int8_first":
            assert st["i8_refined"] == 0 and st["bound_violations"] == 0, st  # ... which is the int8 level: nothing left for the hi pass
        assert 0.0 < st["err_ratio_max"] <= 0.5, st             # the bound has a margin of at least 2x on this corpus
        per = hits.reshape(nq, k)
        sample = [0, 31, 64, 100, 127, 128, 200, 255]
        ex, _ = store.query(queries[sample], Metric.Cosine).take(k).with_path(Path.Exact).per_query().collect_arrays()
        ex = ex.reshape(len(sample), k)
        for j, qi in enumerate(sample):
            assert np.array_equal(per[qi]["index"], ex[j]["index"]), (mode, qi)
            assert np.array_equal(per[qi]["score"].view(np.uint32), ex[j]["score"].view(np.uint32)), (mode, qi)
            assert np.all(per[qi]["query"] == qi)
        for qi in (7, 250):  # every hit re-derived by the oracle
            sc = _oracle_scores(oracle, store, per[qi]["index"], queries[qi], dim)
            assert np.array_equal(per[qi]["score"].view(np.uint32), sc.view(np.uint32))
            assert np.all(np.diff(per[qi]["score"]) <= 0)
        # sampled completeness for four queries: nothing in two 50k-row windows beats a query's k-th score unless it is listed
        rng = np.random.default_rng(1)
        for start in rng.integers(0, n - 50_000, 2):
            blk = oracle.rand_rows(int(start), 50_000, dim, SEED)
            for qi in (3, 90, 180, 254):
                s = oracle.vec_query(blk, queries[qi], oracle.METRIC_COSINE, oracle.TAKE_MAX, 1, fast=True)
                assert s["score"][0] <= per[qi]["score"][-1] or (int(s["index"][0]) + int(start)) in set(per[qi]["index"].tolist())
        # merged = the reference's semantics (src/vec.rs:217-219): the canonical merge of the per-query lists
        m, _ = store.query(queries, Metric.Cosine).take(k).with_path(Path.Mfma).collect_arrays()
        flat = hits[np.lexsort((hits["query"], hits["index"], -hits["score"].astype(np.float64)))][:k]
        assert np.array_equal(m["index"], flat["index"]) and np.array_equal(m["query"], flat["query"])
        assert np.array_equal(m["score"].view(np.uint32), flat["score"].view(np.uint32))
    finally:
        for name in opts:
            store.set_option(name, -1 if name == "hi_fmt" else 0)
        if "hi_fmt" in opts:
            store.set_batch_image(False)
            store.set_batch_image(True)


def test_certification_margin_over_fuzz_corpus(oracle):
    """How much of the error bound each candidate pass really uses: max |approximate - exact| / eps over batches of varied
    shape, magnitude and metric.  The f32-pipe and split-bf16 bounds rest on an accumulation-error model of the MFMA
     that AMD does not document, so their margin is measured, not assumed: it must stay below 0.5 (observed
    ~0.1 and ~0.17).  The hi pass's bound is dominated by the MEASURED operand rounding loss combined by Cauchy-Schwarz — a
    theorem, tight when the rounding errors happen to line up with the other operand (small dims) — so there the
    requirement is the bound itself, <= 1 (observed up to ~0.75 at dim 8, < 0.5 at dim 768); both element formats of the hi
    plane (IEEE half, the default, and bf16) are held to it."""
    worst = {}
    for seed in range(12):
        rng = np.random.default_rng(900 + seed)
        n, dim, nq = int(rng.integers(3000, 60000)), int(rng.choice([8, 33, 96, 200, 768])), int(rng.choice([6, 24, 70, 140, 260]))
        scale = np.exp(rng.normal(0, 1.0, (n, 1))) if seed % 2 else 1.0
        rows = (rng.normal(0, 1, (n, dim)) * scale).astype(np.float32) if seed % 3 else rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        queries = rng.normal(0, 1, (nq, dim)).astype(np.float32)
        for mode, opts in (("hi", {}), ("hi_bf16", {"hi_fmt": 0}), ("split", {"no_hi_pass": 1}), ("f32", {"mfma_f32": 1})):
            store = VecStore(dim)
            for name, v in opts.items():
                store.set_option(name, v)
            store.add_vectors(rows)
            for metric in (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
                a, ca = store.query(queries, metric).take(10).with_path(Path.Mfma).per_query().collect_arrays()
                st = store.last_stats
                assert st["path_used"] == 2
                worst[mode] = max(worst.get(mode, 0.0), st["err_ratio_max"])
                b, cb = store.query(queries, metric).take(10).with_path(Path.Exact).per_query().collect_arrays()
                assert ca == cb and np.array_equal(a["index"], b["index"]) and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))
    print("max |approx - exact| / eps per pass:", worst)
    assert set(worst) == {"hi", "hi_bf16", "split", "f32"}
    assert 0.0 < worst["f32"] <= 0.5 and 0.0 < worst["split"] <= 0.5 and 0.0 < worst["hi"] <= 1.0 and 0.0 < worst["hi_bf16"] <= 1.0, worst


# ---- round 3: the other two metrics at full size, wide rows, and config 4's real shard shape -----------------------------------

def _oracle_metric_scores(oracle, store, idx, q, dim, metric, seed=SEED):
    out = []
    for i in idx:
        row = oracle.rand_rows(int(i), 1, dim, seed)[0]
        if metric == Metric.Cosine:
            out.append(oracle.cosine(q, row, oracle.inv_norms(q)[0], oracle.inv_norms(row)[0]))
        elif metric == Metric.Euclidean:
            out.append(oracle.l2sq(q, row))
        else:
            out.append(oracle.dot(q, row))
    return np.array(out, np.float32)


def test_default_take_of_a_256_query_batch_at_full_size(oracle, big):
    """Round 5: the reference's default take is k = n_vecs over the flattened nq x n score matrix (src/vec.rs:213-219) at ANY
    size; `store.query(batch_of_256).collect()` on 10M x 768 is 2.56e9 (row, query) pairs and used to be an error (the sort
    path stopped at 2^31 pairs and would have needed 51 GB of scratch there).  Now the path runs over row slices of 2^29 pairs
    with the gate carried across them (run_large_k): 10M hits come back, best first; the first 500 equal the canonical merge
    of the per-query top-500 lists of the register-list kernel (an independent path); a 1000-hit sample has the oracle's score
    bits; and a sampled completeness check: in two 20k-row windows no (row, query) pair beats the 10M-th score unless it is
    listed."""
    meta, n, dim, cs = big
    store = meta._store
    nq = 256
    queries = oracle.rand_rows(0, nq, dim, SEED + 1)
    hits, counts = store.query(queries, Metric.Cosine).with_path(Path.Exact).collect_arrays()   # default take: k = n rows
    assert hits.size == n and sum(counts) == n
    sc = hits["score"]
    assert np.all(sc[:-1] >= sc[1:])
    pair = hits["index"].astype(np.uint64) * np.uint64(nq) + hits["query"].astype(np.uint64)
    assert np.unique(pair).size == n  # no pair twice
    # the head: per-query top-500 from the register lists, merged canonically (score desc, row asc, query asc)
    pq, cnt = store.query(queries, Metric.Cosine).take(500).per_query().with_path(Path.Exact).collect_arrays()
    assert cnt == [500] * nq
    order = np.lexsort((pq["query"], pq["index"], -pq["score"].astype(np.float64)))[:500]
    head = pq[order]
    assert np.array_equal(hits["index"][:500], head["index"]) and np.array_equal(hits["query"][:500], head["query"])
    assert np.array_equal(hits["score"][:500].view(np.uint32), head["score"].view(np.uint32))
    # a sample of 1000 hits over the whole result: every score re-derived by the oracle from the regenerated row
    rng = np.random.default_rng(5)
    for i in np.sort(rng.choice(n, 1000, replace=False)):
        row = oracle.rand_rows(int(hits["index"][i]), 1, dim, SEED)[0]
        q = queries[int(hits["query"][i])]
        want = oracle.cosine(q, row, oracle.inv_norms(q)[0], oracle.inv_norms(row)[0])
        assert np.float32(want).view(np.uint32) == hits["score"][i].view(np.uint32), i
    # completeness, sampled: whatever scores above the last hit inside a window of rows is listed
    last = float(sc[-1])
    listed = np.sort(pair)
    for start in rng.integers(0, n - 20_000, 2):
        blk = oracle.rand_rows(int(start), 20_000, dim, SEED)
        inv = oracle.inv_norms(blk)
        for qi in rng.choice(nq, 6, replace=False):
            s_all = oracle.vec_query(blk, queries[qi], oracle.METRIC_COSINE, oracle.TAKE_MAX, 200, inv=inv, fast=True)
            above = s_all[s_all["score"] > last]
            want = (above["index"].astype(np.uint64) + np.uint64(start)) * np.uint64(nq) + np.uint64(qi)
            pos = np.searchsorted(listed, want)
            assert np.all(pos < listed.size) and np.array_equal(listed[np.minimum(pos, listed.size - 1)], want), (start, qi)
    # take(1000) merged over 3M rows x 1024 queries (3.07e9 pairs): slow no longer means an error either
    q1024 = oracle.rand_rows(0, 1024, dim, SEED + 7)
    cmask = np.zeros((n + cs - 1) // cs, dtype=bool)
    cmask[: 3_000_000 // cs] = True
    m, _ = store._run(store.query(q1024, Metric.Cosine).take(1000).with_path(Path.Exact).resolve(), chunk_mask=cmask)[:2]
    assert m.size == 1000 and np.all(m["score"][:-1] >= m["score"][1:]) and int(m["index"].max()) < 3_000_000 // cs * cs
    p2, c2 = store._run(store.query(q1024[:64], Metric.Cosine).take(100).per_query().with_path(Path.Exact).resolve(), chunk_mask=cmask)[:2]
    # (the best 100 of the first 64 queries through the register lists: the merged top-1000's entries of those queries that
    #  score at least as well as a query's 100-th must all be among them)
    for qi in range(64):
        mine = m[m["query"] == qi]
        lst = p2[qi * 100:(qi + 1) * 100]
        inside = mine[mine["score"] >= lst["score"][-1]]
        assert np.all(np.isin(inside["index"], lst["index"])), qi


@pytest.mark.parametrize("metric", [Metric.Euclidean, Metric.DotProduct], ids=["euclidean", "dot"])
def test_config2_shape_euclidean_and_dot(oracle, big, metric):
    """BASELINE config 2's shape (10M x 768, 256 queries, take(100)) for the two metrics the cosine tests above do not cover
    (src/vec_compute.rs:9-22 dot, :35-54 squared L2 — the metric whose candidate pass goes through the cancellation-prone
    expansion |q|^2 + |v|^2 - 2 q.v).  Default cascade; take_min for L2 (the reference's inferred default, src/vec.rs:92-98).
    No query may fall through, 8 sampled queries equal the exact-order kernel bit for bit, every hit of two queries is
    re-derived by the oracle, sampled completeness, and the certification bound is used to less than half."""
    meta, n, dim, cs = big
    store = meta._store
    nq, k = 256, 100
    queries = oracle.rand_rows(0, nq, dim, SEED + 1)
    tmax = metric != Metric.Euclidean
    o_metric, o_take = int(metric), (oracle.TAKE_MAX if tmax else oracle.TAKE_MIN)
    hits, counts = store.query(queries, metric).take(k).with_path(Path.Mfma).per_query().collect_arrays()
    st = dict(store.last_stats)
    assert st["path_used"] == 2 and counts == [k] * nq
    assert st["refined"] == 0 and st["retries"] == 0 and st["gate_failed"] == 0 and st["bound_violations"] == 0, st
    assert 0.0 < st["err_ratio_max"] <= 0.5, st
    per = hits.reshape(nq, k)
    sample = [0, 31, 64, 100, 127, 128, 200, 255]
    ex, _ = store.query(queries[sample], metric).take(k).with_path(Path.Exact).per_query().collect_arrays()
    ex = ex.reshape(len(sample), k)
    for j, qi in enumerate(sample):
        assert np.array_equal(per[qi]["index"], ex[j]["index"]), (metric, qi)
        assert np.array_equal(per[qi]["score"].view(np.uint32), ex[j]["score"].view(np.uint32)), (metric, qi)
    for qi in (7, 250):
        sc = _oracle_metric_scores(oracle, store, per[qi]["index"], queries[qi], dim, metric)
        assert np.array_equal(per[qi]["score"].view(np.uint32), sc.view(np.uint32))
        d = np.diff(per[qi]["score"])
        assert np.all(d <= 0) if tmax else np.all(d >= 0)
    rng = np.random.default_rng(5)
    for start in rng.integers(0, n - 50_000, 2):
        blk = oracle.rand_rows(int(start), 50_000, dim, SEED)
        for qi in (3, 180):
            s = oracle.vec_query(blk, queries[qi], o_metric, o_take, 1, fast=True)
            kth = per[qi]["score"][-1]
            beaten = s["score"][0] > kth if tmax else s["score"][0] < kth
            assert not beaten or (int(s["index"][0]) + int(start)) in set(per[qi]["index"].tolist())
    # merged (the reference's semantics) = canonical merge of the per-query lists
    m, _ = store.query(queries, metric).take(k).with_path(Path.Mfma).collect_arrays()
    key = -hits["score"].astype(np.float64) if tmax else hits["score"].astype(np.float64)
    flat = hits[np.lexsort((hits["query"], hits["index"], key))][:k]
    assert np.array_equal(m["index"], flat["index"]) and np.array_equal(m["query"], flat["query"])
    assert np.array_equal(m["score"].view(np.uint32), flat["score"].view(np.uint32))


@pytest.mark.parametrize("metric", [Metric.Euclidean, Metric.DotProduct], ids=["euclidean", "dot"])
def test_headline_shape_euclidean_and_dot(oracle, big, metric):
    """The headline shape (10M x 768, ONE query, take(10)) for squared L2 and dot: exact-order kernel == the cascade (single
    query over the resident bf16 plane), every score re-derived by the oracle, sampled completeness."""
    meta, n, dim, cs = big
    store = meta._store
    tmax = metric != Metric.Euclidean
    q = oracle.rand_rows(0, 1, dim, SEED + 7)[0]
    a, _ = store.query(q, metric).take(10).with_path(Path.Exact).collect_arrays()
    assert store.last_stats["path_used"] == 1 and a.size == 10
    b, _ = store.query(q, metric).take(10).with_path(Path.Mfma).collect_arrays()
    stb = dict(store.last_stats)
    assert stb["path_used"] == 2 and stb["retries"] == 0
    assert np.array_equal(a["index"], b["index"]) and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))
    sc = _oracle_metric_scores(oracle, store, a["index"], q, dim, metric)
    assert np.array_equal(a["score"].view(np.uint32), sc.view(np.uint32))
    rng = np.random.default_rng(11)
    for start in rng.integers(0, n - 50_000, 3):
        blk = oracle.rand_rows(int(start), 50_000, dim, SEED)
        s = oracle.vec_query(blk, q, int(metric), oracle.TAKE_MAX if tmax else oracle.TAKE_MIN, 1, fast=True)
        beaten = s["score"][0] > a["score"][-1] if tmax else s["score"][0] < a["score"][-1]
        assert not beaten or (int(s["index"][0]) + int(start)) in set(a["index"].tolist())


@pytest.mark.parametrize("dim,n", [(1536, 1_200_000), (3072, 1_000_000)], ids=["dim1536", "dim3072"])
def test_wide_rows_1536_and_3072(oracle, dim, n):
    """Common embedding widths past the in-argument query limit (896 floats: the single query is uploaded instead of riding
    in the kernel arguments) and past one 2048-float LDS query block: >= 1M rows, 1 and 64 queries, all three metrics.
    Exact path == batch cascade bit for bit, every hit of the single query and of two batch queries re-derived by the
    oracle, and the single-query top-10 over the first 60k rows equals the oracle's outright."""
    seed = SEED + dim
    store = VecStore(dim)
    store.append_random(n, seed)
    Q = oracle.rand_rows(0, 64, dim, seed + 1)
    head = oracle.rand_rows(0, 60_000, dim, seed)
    try:
        for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
            tmax = metric != Metric.Euclidean
            a, _ = store.query(Q[0], metric).take(10).with_path(Path.Exact).collect_arrays()
            assert store.last_stats["path_used"] == 1 and a.size == 10
            sc = _oracle_metric_scores(oracle, store, a["index"], Q[0], dim, metric, seed)
            assert np.array_equal(a["score"].view(np.uint32), sc.view(np.uint32)), (dim, metric)
            # the first 60k rows alone (chunk mask: 1024-row chunks) against the oracle's top-10 over them
            cm = np.zeros((n + 1023) // 1024, bool)
            cm[: 60_000 // 1024] = True
            nh = (60_000 // 1024) * 1024
            rq = store.query(Q[0], metric).take(10).with_path(Path.Exact).resolve()
            h, _, _ = store._run(rq, chunk_mask=cm)
            ref = oracle.vec_query(head[:nh], Q[0], int(metric), oracle.TAKE_MAX if tmax else oracle.TAKE_MIN, 10, ties=oracle.TIES_CANONICAL, fast=True)
            assert np.array_equal(h["index"], ref["index"]) and np.array_equal(h["score"].view(np.uint32), ref["score"].view(np.uint32))
            # 64 queries: cascade == exact path, per query
            b, cb = store.query(Q, metric).take(10).with_path(Path.Mfma).per_query().collect_arrays()
            stb = dict(store.last_stats)
            assert stb["path_used"] == 2 and cb == [10] * 64 and stb["retries"] == 0, stb
            e, ce = store.query(Q, metric).take(10).with_path(Path.Exact).per_query().collect_arrays()
            assert np.array_equal(b["index"], e["index"]) and np.array_equal(b["score"].view(np.uint32), e["score"].view(np.uint32))
            assert np.array_equal(b[:10]["index"], a["index"])  # query 0 of the batch == the single query
            for qi in (17, 63):
                sc = _oracle_metric_scores(oracle, store, b[qi * 10:(qi + 1) * 10]["index"], Q[qi], dim, metric, seed)
                assert np.array_equal(b[qi * 10:(qi + 1) * 10]["score"].view(np.uint32), sc.view(np.uint32))
    finally:
        store.close()


@pytest.fixture(scope="module")
def c4_shard():
    """One shard of BASELINE config 4 at its real size: 5M x 768 rows = 15.36 GB, as rank 3 of 8 would hold it (global rows
    15M .. 20M: hits must carry global indices, src/meta_compute.rs:185)."""
    n, dim, base = 5_000_000, 768, 15_000_000
    store = VecStore(dim)
    store.set_base_offset(base)
    store.append_random(n, SEED)
    yield store, n, dim, base
    store.close()


@pytest.mark.parametrize("coop", [1, 0], ids=["siblings", "one_workgroup_per_tile"])
def test_config4_real_shard_shape_1024_queries_top100(oracle, c4_shard, coop):
    """BASELINE config 4's per-GPU work at its real shape: 5M x 768 rows, a 1024-query batch, cosine, take(100), per query
    (what the shard contributes to the all-gather) and merged (the reference's semantics, src/vec.rs:217-219); with the
    four 256-query blocks of a row tile on sibling workgroups of one XCD (`mfma_coop`, the default) and one after the other
    on one workgroup.  Certified by the first pass for all 1024 queries; 8 sampled queries equal the exact-order kernel bit
    for bit; every hit of two queries is re-derived by the oracle from the regenerated row; sampled completeness."""
    store, n, dim, base = c4_shard
    nq, k = 1024, 100
    queries = oracle.rand_rows(0, nq, dim, SEED + 4)
    store.set_option("force_fallback", 0 if coop else 4)  # bit 4: the blocks of a row tile one after the other on one workgroup
    try:
        hits, counts = store.query(queries, Metric.Cosine).take(k).with_path(Path.Mfma).per_query().collect_arrays()
        st = dict(store.last_stats)
        assert st["path_used"] == 2 and counts == [k] * nq
        assert st["refined"] == 0 and st["retries"] == 0 and st["gate_failed"] == 0 and st["bound_violations"] == 0, st
        assert 0.0 < st["err_ratio_max"] <= 0.5, st
        assert st["vectors_compared"] == n * nq
        per = hits.reshape(nq, k)
        assert per["index"].min() >= base and per["index"].max() < base + n
        sample = [0, 255, 256, 511, 512, 700, 1000, 1023]  # every 256-query block, both ends
        ex, _ = store.query(queries[sample], Metric.Cosine).take(k).with_path(Path.Exact).per_query().collect_arrays()
        ex = ex.reshape(len(sample), k)
        for j, qi in enumerate(sample):
            assert np.array_equal(per[qi]["index"], ex[j]["index"]), qi
            assert np.array_equal(per[qi]["score"].view(np.uint32), ex[j]["score"].view(np.uint32)), qi
            assert np.all(per[qi]["query"] == qi)
        for qi in (300, 900):
            sc = []
            for i in per[qi]["index"]:
                row = oracle.rand_rows(int(i), 1, dim, SEED)[0]  # the generator is keyed by the GLOBAL row
                sc.append(oracle.cosine(queries[qi], row, oracle.inv_norms(queries[qi])[0], oracle.inv_norms(row)[0]))
            assert np.array_equal(per[qi]["score"].view(np.uint32), np.array(sc, np.float32).view(np.uint32))
            assert np.all(np.diff(per[qi]["score"]) <= 0)
        rng = np.random.default_rng(2)
        for start in rng.integers(0, n - 50_000, 2):
            blk = oracle.rand_rows(base + int(start), 50_000, dim, SEED)
            for qi in (1, 513, 1022):
                s = oracle.vec_query(blk, queries[qi], oracle.METRIC_COSINE, oracle.TAKE_MAX, 1, fast=True)
                assert s["score"][0] <= per[qi]["score"][-1] or (int(s["index"][0]) + base + int(start)) in set(per[qi]["index"].tolist())
        m, _ = store.query(queries, Metric.Cosine).take(k).with_path(Path.Mfma).collect_arrays()
        flat = hits[np.lexsort((hits["query"], hits["index"], -hits["score"].astype(np.float64)))][:k]
        assert np.array_equal(m["index"], flat["index"]) and np.array_equal(m["query"], flat["query"])
        assert np.array_equal(m["score"].view(np.uint32), flat["score"].view(np.uint32))
    finally:
        store.set_option("force_fallback", 0)


def test_config4_whole_corpus_in_one_process_eight_shards(oracle):
    """BASELINE config 4 AS STATED except for the number of physical devices: 40M x 768 f32 (122.9 GB of rows + 61 GB of 16-bit
    planes, all on this box's one 288-GB GPU), eight shards of 5M rows behind ONE store of ONE process (ott_store_create_multi),
    a 1024-query batch, cosine, take(100) per query through the unchanged call: every shard runs the matrix-core cascade, the
    [1024][128]-slot blocks are exchanged and merged on the first shard's GPU.  Certified by the first pass everywhere; sampled
    queries equal the exact-order path over all 40M rows bit for bit; every hit of two queries re-derived by the oracle from
    the regenerated GLOBAL row; sampled completeness; the merged form (the reference's semantics) equals the flattened lists."""
    import ctypes as C
    from otters_amd import _native as N
    free, total = C.c_size_t(0), C.c_size_t(0)
    N.lib()
    hip = C.CDLL(None)
    if hip.hipMemGetInfo(C.byref(free), C.byref(total)) != 0 or free.value < 188 * 2 ** 30:
        pytest.skip("needs ~190 GB of free HBM")
    n, dim, nq, k, shards = 40_000_000, 768, 1024, 100, 8
    store = VecStore(dim, devices=[0] * shards)
    store.reserve(n)
    store.append_random(n, SEED)
    layout = store.shards()
    assert len(layout) == shards and sum(c for _, _, c in layout) == n and all(abs(c - n // shards) <= 1024 for _, _, c in layout)
    queries = oracle.rand_rows(0, nq, dim, SEED + 4)
    hits, counts = store.query(queries, Metric.Cosine).take(k).with_path(Path.Mfma).per_query().collect_arrays()
    st = dict(store.last_stats)
    assert st["path_used"] == 2 and counts == [k] * nq
    assert st["refined"] == 0 and st["retries"] == 0 and st["gate_failed"] == 0 and st["bound_violations"] == 0, st
    assert st["vectors_compared"] == n * nq and st["total_chunks"] == (n + 1023) // 1024
    per = hits.reshape(nq, k)
    assert per["index"].max() < n and len({int(i) // (n // shards) for i in per[5]["index"]}) > 1  # a query's hits come from several shards
    sample = [0, 511, 1023]
    ex, _ = store.query(queries[sample], Metric.Cosine).take(k).with_path(Path.Exact).per_query().collect_arrays()
    ex = ex.reshape(len(sample), k)
    for j, qi in enumerate(sample):
        assert np.array_equal(per[qi]["index"], ex[j]["index"]), qi
        assert np.array_equal(per[qi]["score"].view(np.uint32), ex[j]["score"].view(np.uint32)), qi
        assert np.all(per[qi]["query"] == qi)
    for qi in (300, 900):
        sc = []
        for i in per[qi]["index"]:
            row = oracle.rand_rows(int(i), 1, dim, SEED)[0]
            sc.append(oracle.cosine(queries[qi], row, oracle.inv_norms(queries[qi])[0], oracle.inv_norms(row)[0]))
        assert np.array_equal(per[qi]["score"].view(np.uint32), np.array(sc, np.float32).view(np.uint32))
        assert np.all(np.diff(per[qi]["score"]) <= 0)
    rng = np.random.default_rng(4)
    for start in rng.integers(0, n - 50_000, 2):
        blk = oracle.rand_rows(int(start), 50_000, dim, SEED)
        for qi in (1, 1022):
            s1 = oracle.vec_query(blk, queries[qi], oracle.METRIC_COSINE, oracle.TAKE_MAX, 1, fast=True)
            assert s1["score"][0] <= per[qi]["score"][-1] or (int(s1["index"][0]) + int(start)) in set(per[qi]["index"].tolist())
    m, _ = store.query(queries, Metric.Cosine).take(k).with_path(Path.Mfma).collect_arrays()
    flat = hits[np.lexsort((hits["query"], hits["index"], -hits["score"].astype(np.float64)))][:k]
    assert np.array_equal(m["index"], flat["index"]) and np.array_equal(m["query"], flat["query"])
    assert np.array_equal(m["score"].view(np.uint32), flat["score"].view(np.uint32))
    store.close()


def test_certification_margin_on_adversarial_sums(oracle):
    """The accumulation-error terms of the certification bounds ((1.25 / 2.5 / 3.75) x dim x 2^-24 relative to
    sum |q_i v_i|) model the matrix unit's summation, which AMD does not document.  Symmetric random data is kind to any
    summation order (errors cancel); the unkind case is what this test feeds: every product positive and of similar size (rows
    and queries uniform in [0.5, 1)), so rounding errors cannot cancel and the sum's magnitude is the sum of magnitudes, at the
    longest supported rows (dim 3072: 3072-term sums) and a short one.  The measured |approximate - exact| / eps of every pass
    must stay below 1 — observed well below — and the results must still be the exact path's bit for bit."""
    rng = np.random.default_rng(4242)
    worst = {}
    for dim, n in ((3072, 20_000), (768, 60_000), (72, 100_000)):
        rows = rng.uniform(0.5, 1.0, (n, dim)).astype(np.float32)
        queries = rng.uniform(0.5, 1.0, (24, dim)).astype(np.float32)
        for mode, opts in (("hi", {}), ("hi_bf16", {"hi_fmt": 0}), ("split", {"no_hi_pass": 1}), ("f32", {"mfma_f32": 1})):
            store = VecStore(dim)
            for name, v in opts.items():
                store.set_option(name, v)
            store.add_vectors(rows)
            for metric in (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
                a, ca = store.query(queries, metric).take(10).with_path(Path.Mfma).per_query().collect_arrays()
                st = store.last_stats
                assert st["path_used"] == 2
                worst[(mode, dim)] = max(worst.get((mode, dim), 0.0), st["err_ratio_max"])
                b, cb = store.query(queries, metric).take(10).with_path(Path.Exact).per_query().collect_arrays()
                assert ca == cb and np.array_equal(a["index"], b["index"]) and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)), (mode, dim, metric)
            store.close()
    print("adversarial sums, max |approx - exact| / eps:", {f"{m}@{d}": round(v, 3) for (m, d), v in worst.items()})
    assert all(0.0 <= v <= 1.0 for v in worst.values()), worst
    assert max(v for (m, _), v in worst.items() if m in ("split", "f32")) <= 0.75, worst


def test_store_filling_most_of_the_gpus_memory():
    """Memory laid out for 288 GB: a store of ~70 % of what is free right now (215 GB on an idle MI355X; at least 100 GB or the
    test is skipped), generated on the device, in a process of its own.  Near-duplicates of the query planted over the whole
    store — the last row included: byte offsets beyond 2^37 — come back first with the oracle's score bits for all three
    metrics; with no room left for the 16-bit plane the batch path runs its split-bf16 pass on the f32 rows, and 8- and
    64-query batches through it equal the exact-order path bit for bit (benchmarks/big_store.py)."""
    import subprocess
    import sys
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    n = int(free * 0.7 / (768 * 4))
    if n < 33_000_000:
        pytest.skip(f"only {free / 1e9:.0f} GB of device memory free")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "benchmarks", "big_store.py"), str(n), "768"], capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0 and "BIG STORE OK" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


def test_growth_drops_the_planes_when_the_new_rows_do_not_fit_next_to_them():
    """Round 5 (advisor): the batch path's copies of the corpus (the int8 plane, built in the background after appends; the half
    plane once a query needed it) must not make a growing store fail — a reallocation drops them anyway, so when the new buffers
    do not fit NEXT TO them they go first and the allocation is tried again.  Sized from what is free on the GPU right now: rows
    = 30 % of it, + int8 plane (7.5 %) + half plane (15 %); the reserve asks for 62 % more: 114 % with the planes, 92 % without."""
    import torch
    free_b = torch.cuda.mem_get_info(0)[0]
    dim = 768
    n = int(0.30 * free_b / (dim * 4 + 5)) // 4096 * 4096
    n_big = int(0.62 * free_b / (dim * 4 + 5)) // 4096 * 4096
    if n < 1_000_000:
        pytest.skip("too little free memory to show anything")
    store = VecStore(dim)
    store.reserve(n)  # (exact capacity: a store that grows by itself doubles, and its planes are sized by the capacity)
    store.append_random(n, 3)
    rng = np.random.default_rng(1)
    q = rng.uniform(-1, 1, (8, dim)).astype(np.float32)
    a, _ = store.query(q, Metric.Cosine).take(10).with_path(Path.Mfma).collect_arrays()      # builds the int8 plane (if the builder has not)
    b, _ = store.query(q, Metric.Euclidean).take(10).with_path(Path.Mfma).collect_arrays()   # squared L2: builds the half plane
    assert store.last_stats["path_used"] == 2
    store.reserve(n_big)
    assert store.len() == n
    a2, _ = store.query(q, Metric.Cosine).take(10).with_path(Path.Mfma).collect_arrays()     # the int8 plane is built again at the new capacity
    b2, _ = store.query(q, Metric.Euclidean).take(10).with_path(Path.Mfma).collect_arrays()  # (no room for the half plane now: split pass)
    for x, y in ((a, a2), (b, b2)):
        assert np.array_equal(x["index"], y["index"]) and np.array_equal(x["score"].view(np.uint32), y["score"].view(np.uint32))
    e, _ = store.query(q[:2], Metric.Cosine).take(10).with_path(Path.Exact).collect_arrays()
    m, _ = store.query(q[:2], Metric.Cosine).take(10).with_path(Path.Mfma).collect_arrays()
    assert np.array_equal(e["index"], m["index"]) and np.array_equal(e["score"].view(np.uint32), m["score"].view(np.uint32))
    store.close()
