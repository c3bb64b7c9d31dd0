"""GPU parity tests for the MFMA batch path: scores come off the matrix cores in a different
summation order, so the path re-scores its candidates in the reference's order and certifies
the result; what it returns must equal the oracle bit for bit, like the exact path."""
import numpy as np
import pytest

from helpers import oracle_collect, same_modulo_ties
from otters_amd import Cmp, Metric, Path, VecStore

pytestmark = pytest.mark.gpu


def run(plan):
    rq = plan.resolve()
    hits, counts, stats = plan.vector_store._run(rq)
    return rq, hits, counts, stats


def assert_bit_exact(hits, ref):
    assert hits.shape == ref.shape, (hits.shape, ref.shape)
    assert np.array_equal(hits["index"], ref["index"]), (hits[:8], ref[:8])
    assert np.array_equal(hits["score"].view(np.uint32), ref["score"].view(np.uint32)), (hits[:8], ref[:8])
    assert np.array_equal(hits["query"], ref["query"])


CASES = [  # (n, dim, nq)
    (300, 8, 3), (1000, 37, 40), (5000, 64, 70), (20000, 128, 256), (9000, 768, 33), (3000, 100, 300), (70000, 96, 64),
]


@pytest.mark.parametrize("shape", CASES, ids=lambda s: "n%d_d%d_q%d" % s)
@pytest.mark.parametrize("metric", [Metric.Cosine, Metric.DotProduct, Metric.Euclidean], ids=lambda m: m.name)
def test_mfma_matches_oracle(oracle, shape, metric):
    n, dim, nq = shape
    rng = np.random.default_rng(n + dim + nq)
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    for k, kind in ((10, "take"), (100, "take"), (25, "take_min")):
        plan = getattr(store.query(queries, metric), kind)(k).with_path(Path.Mfma)
        rq, hits, _, stats = run(plan)
        assert stats["path_used"] == 2
        assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
        # per-query lists
        plan = getattr(store.query(queries, metric), kind)(k).with_path(Path.Mfma).per_query()
        rq, hits, counts, stats = run(plan)
        o = 0
        for qi in range(0, nq, max(nq // 7, 1)):
            ref = oracle.vec_query(rows, queries[qi], rq.metric, rq.take, k, ties=oracle.TIES_CANONICAL)
            start = sum(counts[:qi])
            got = hits[start:start + counts[qi]]
            assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
            assert (got["query"] == qi).all()


def test_mfma_filter_and_masks(oracle):
    rng = np.random.default_rng(3)
    n, dim, nq, cs = 30000, 48, 50, 700
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    store = VecStore(dim)
    store.set_chunk_size(cs)
    store.add_vectors(rows)
    row_mask = rng.random(n) < 0.6
    chunk_mask = rng.random((n + cs - 1) // cs) < 0.5
    for thr, cmp in ((0.3, Cmp.Gt), (0.25, Cmp.Gte), (0.0, Cmp.Lt), (0.31, Cmp.Lte)):
        plan = store.query(queries, Metric.Cosine).filter(thr, cmp).with_row_mask(row_mask).take(40).with_path(Path.Mfma)
        rq = plan.resolve()
        hits, _, stats = store._run(rq, chunk_mask=chunk_mask)
        ref, rstats = oracle.meta_query(rows, cs, queries, rq.metric, rq.take, rq.k, rq.filter_cmp, rq.filter_thr,
                                        chunk_mask=chunk_mask, row_mask=row_mask, ties=oracle.TIES_CANONICAL)
        assert_bit_exact(hits, ref)
        assert stats["vectors_compared"] == rstats["vectors_compared"]


def test_mfma_ties_fall_back_to_exact(oracle):
    # heavy exact ties (quantised data + duplicates): certification must refuse and the exact path must answer
    rng = np.random.default_rng(9)
    rows = rng.integers(-2, 3, (8000, 16)).astype(np.float32)
    queries = rng.integers(-2, 3, (40, 16)).astype(np.float32)
    store = VecStore(16)
    store.add_vectors(rows)
    for metric in (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
        plan = store.query(queries, metric).take(20).with_path(Path.Mfma)
        rq, hits, _, stats = run(plan)
        assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
        lit = oracle_collect(oracle, rq, rows, oracle.TIES_LITERAL)
        same_modulo_ties(hits["index"], hits["score"], lit["index"], lit["score"], hits["query"], lit["query"])


def test_auto_path_picks_mfma_for_big_batches(oracle):
    store = VecStore(64)
    store.append_random(70000, seed=5)
    rows = oracle.rand_rows(0, 70000, 64, 5)
    q = np.random.default_rng(1).uniform(-1, 1, (64, 64)).astype(np.float32)
    rq, hits, _, stats = run(store.query(q, Metric.Cosine).take(10))
    assert stats["path_used"] == 2
    assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
    rq, hits, _, stats = run(store.query(q[:4], Metric.Cosine).take(10))
    assert stats["path_used"] == 1


def test_mfma_euclidean_near_duplicates(oracle):
    """squared-L2 on the matrix cores goes through ||q||^2 + ||v||^2 - 2 q.v, which cancels for near-duplicates:
    the error bound must catch that (certify or fall back) and the answer must still be the oracle's"""
    rng = np.random.default_rng(4)
    n, dim, nq = 40000, 64, 48
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    for i in range(nq):  # a few rows extremely close to each query
        for j in range(3):
            rows[800 * i + 17 * j + 5] = queries[i] + rng.normal(0, 1e-4, dim).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    plan = store.query(queries, Metric.Euclidean).take(10).with_path(Path.Mfma)
    rq, hits, _, stats = run(plan)
    assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
    plan = store.query(queries, Metric.Euclidean).filter(50.0, Cmp.Lt).take(64).with_path(Path.Mfma)
    rq, hits, _, stats = run(plan)
    assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))


OPERAND_MODES = {"f32_pipe": {"OTT_MFMA_F32": "1"}, "split_in_registers": {"OTT_NO_BATCH_IMAGE": "1"},
                 "batch_image": {"OTT_NO_HI_PASS": "1"}, "hi_pass_cascade": {},
                 # the cascade's fallback paths forced on (option force_fallback: sequential query blocks, the open first round
                 # through the cursor atomics, conservative gates)
                 "hi_pass_cascade_fallbacks": {"OTT_FORCE_FALLBACK": str(4 + 16 + 32)},
                 # round 5: the int8 plane as the cascade's first level (cosine / dot; squared L2 starts at the hi pass as before)
                 "int8_first": {"OTT_HI_FMT": "2"}}


@pytest.mark.parametrize("mode", list(OPERAND_MODES), ids=list(OPERAND_MODES))
def test_batch_operand_modes_agree_with_oracle(oracle, mode, monkeypatch):
    """The candidate pass has four operand modes — f32 matrix pipe, split bf16 with the rows split in registers, split bf16
    from the store's pre-split batch image, and the default cascade (bf16 hi plane first, split pass for what it cannot
    certify) — and all of them must return the oracle's result bit for bit: every tile width,
    every metric, filters, masks, appended and rewritten rows (the image has to follow both), awkward magnitudes."""
    for k in ("OTT_MFMA_F32", "OTT_NO_BATCH_IMAGE", "OTT_NO_HI_PASS", "OTT_FORCE_FALLBACK", "OTT_HI_FMT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in OPERAND_MODES[mode].items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(91)
    n, dim = 30_000, 200
    rows = (rng.normal(0, 1, (n, dim)) * np.exp(rng.normal(0, 1.5, (n, 1)))).astype(np.float32)  # norms over several decades
    store = VecStore(dim)
    store.set_chunk_size(512)
    store.add_vectors(rows[:20_000])
    for nq in (7, 20, 40, 100, 260):
        queries = rng.normal(0, 1, (nq, dim)).astype(np.float32)
        if nq == 40:  # grow the store and rewrite a few rows between batches
            store.add_vectors(rows[20_000:])
            rows[123] = queries[3] * 3.0
            rows[25_000] = -queries[5]
            store.write_rows(123, rows[123:124])
            store.write_rows(25_000, rows[25_000:25_001])
        cur = rows[: store.len()]
        for metric in (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
            plan = store.query(queries, metric).take(30).with_path(Path.Mfma)
            rq, hits, _, stats = run(plan)
            assert stats["path_used"] == 2
            assert_bit_exact(hits, oracle_collect(oracle, rq, cur, oracle.TIES_CANONICAL))
        mask = rng.random(store.len()) < 0.5
        plan = store.query(queries, Metric.Cosine).with_row_mask(mask).filter(0.05, Cmp.Gt).take_min(40).with_path(Path.Mfma)
        rq, hits, _, _ = run(plan)
        assert_bit_exact(hits, oracle_collect(oracle, rq, cur, oracle.TIES_CANONICAL))
        if nq == 100:  # dropping the image mid-way (and allowing it again) changes nothing
            store.set_batch_image(False)
            rq, hits2, _, _ = run(plan)
            assert_bit_exact(hits2, hits)
            store.set_batch_image(True)


def test_cascade_falls_through_and_backs_off(oracle, monkeypatch):
    """Near-duplicate clusters: the k-th best scores of every query sit within 1e-6 of each other, far inside the hi pass's
    bound, so it certifies nothing — every query must fall through (split pass, then exact path) and still come back
    bit-exact; the store first widens the hi pass (512 re-scored candidates per query, what rescues ordinary clustered
    corpora), and when that fails too it skips the hi pass for a while.  A well-separated batch certifies in the hi pass."""
    for k in ("OTT_MFMA_F32", "OTT_NO_BATCH_IMAGE", "OTT_NO_HI_PASS"):
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(17)
    n, dim, nq = 20_000, 128, 16
    queries = rng.normal(0, 1, (nq, dim)).astype(np.float32)
    rows = (queries[rng.integers(0, nq, n)] + rng.normal(0, 1e-3, (n, dim))).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    plan = store.query(queries, Metric.Cosine).take(10).with_path(Path.Mfma)
    rq, hits, _, stats = run(plan)
    ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
    assert_bit_exact(hits, ref)
    assert stats["refined"] == nq, stats   # the hi pass certified nothing
    rq, hits, _, stats2 = run(plan)
    assert_bit_exact(hits, ref)
    assert stats2["refined"] == nq, stats2  # the store's first answer: the hi pass once more, re-scoring 512 per query — no better here
    rq, hits, _, stats3 = run(plan)
    assert_bit_exact(hits, ref)
    assert stats3["refined"] == 0, stats3  # backing off: this batch went straight to the split pass
    # separated data on a fresh store: certified by the hi pass alone
    rows2 = rng.normal(0, 1, (n, dim)).astype(np.float32)
    store2 = VecStore(dim)
    store2.add_vectors(rows2)
    plan2 = store2.query(queries, Metric.Cosine).take(10).with_path(Path.Mfma)
    rq2, hits2, _, stats4 = run(plan2)
    assert_bit_exact(hits2, oracle_collect(oracle, rq2, rows2, oracle.TIES_CANONICAL))
    assert stats4["refined"] == 0 and stats4["retries"] == 0, stats4


def test_cascade_wide_level_resolves_near_duplicate_clusters(oracle, monkeypatch):
    """Clusters of 600 near-identical rows: more rows than the split pass re-scores (512) sit within ITS bound of every query's
    k-th score, so the first two levels certify nothing; the third level (split pass re-scoring 4096 candidates per query)
    certifies all of them — no query is left to the exact path — and the result is still the oracle's bit for bit."""
    for k in ("OTT_MFMA_F32", "OTT_NO_BATCH_IMAGE", "OTT_NO_HI_PASS"):
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(23)
    dim, nq, per = 128, 16, 600
    centres = rng.normal(0, 1, (50, dim)).astype(np.float32)
    rows = (np.repeat(centres, per, axis=0) + rng.normal(0, 1e-4, (50 * per, dim))).astype(np.float32)
    rows = rows[rng.permutation(rows.shape[0])]
    queries = (centres[:nq] + rng.normal(0, 1e-4, (nq, dim))).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    for metric in (Metric.Cosine, Metric.DotProduct):
        plan = store.query(queries, metric).take(10).with_path(Path.Mfma)
        rq, hits, _, stats = run(plan)
        assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
        assert stats["path_used"] == 2 and stats["retries"] == 0, stats
        if metric == Metric.Cosine:
            assert stats["passes"] >= 3, stats  # hi pass, split pass (512), split pass (4096): all three levels ran


def test_prepare_batch_builds_the_hi_plane_ahead(oracle):
    """prepare_batch() is optional warm-up: same results with or without it, also after appends and on an empty store."""
    rng = np.random.default_rng(61)
    dim = 48
    rows = rng.normal(0, 1, (6000, dim)).astype(np.float32)
    queries = rng.normal(0, 1, (9, dim)).astype(np.float32)
    empty = VecStore(dim)
    empty.prepare_batch()  # nothing resident: a no-op
    store = VecStore(dim)
    store.add_vectors(rows[:4000])
    store.prepare_batch()
    store.add_vectors(rows[4000:])
    store.prepare_batch()
    store.prepare_batch()
    plan = store.query(queries, Metric.Cosine).take(5).with_path(Path.Mfma)
    rq, hits, _, stats = run(plan)
    assert stats["path_used"] == 2 and stats["refined"] == 0
    assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))


@pytest.mark.parametrize("nq", [512, 520, 1030])
def test_query_blocks_on_sibling_workgroups(oracle, nq):
    """Batches of 2 / 4 blocks of 256 queries: the blocks of a row tile run on sibling workgroups of one XCD (`mfma_coop`,
    default) or one after the other on one workgroup (option 0; always for 3 blocks: nq 520 pads to 768).  Both mappings and
    the exact-order path return the same hits bit for bit on a store large enough for full persistent rounds (>= 64 tiles of
    256 rows per round); a slice of the queries is also held to the oracle."""
    rng = np.random.default_rng(1000 + nq)
    n, dim, k = 300_000, 64, 20
    store = VecStore(dim)
    store.append_random(n, 77)
    queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    plan = lambda path: store.query(queries, Metric.Cosine).take(k).per_query().with_path(path)
    _, exact, _, _ = run(plan(Path.Exact))
    # every candidate-pass kernel that takes 256-query blocks: hi pass (default cascade), split bf16 pass, f32 matrix pipe
    for mode in ({}, {"no_hi_pass": 1}, {"mfma_f32": 1}):
        for name, v in mode.items():
            store.set_option(name, v)
        for coop in (1, 0):
            store.set_option("force_fallback", 0 if coop else 4)  # bit 4: the blocks of a row tile one after the other
            _, hits, _, stats = run(plan(Path.Mfma))
            assert stats["path_used"] == 2
            assert_bit_exact(hits, exact)
        for name in mode:
            store.set_option(name, 0)
    rows = oracle.rand_rows(0, n, dim, 77)  # the host twin of append_random
    for q in (0, 255, 256, nq - 1):
        rq = store.query(queries[q], Metric.Cosine).take(k).resolve()
        ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
        got = exact[exact["query"] == q]
        assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))


@pytest.mark.parametrize("metric", [Metric.Cosine, Metric.DotProduct, Metric.Euclidean], ids=lambda m: m.name)
def test_speculative_gate_matches_exact_path(oracle, metric):
    """Between the row rounds the batch path emits against a speculative threshold (select_kernel: the j-th best of the rows
    seen so far, j < T) once the store is large enough for rounds that cover less than an eighth of it.  Whatever the gate
    does, the hits are the exact path's bit for bit — plain, with a score filter, with a row mask, take_min, several k — with
    the option on (default) and off; a slice of the queries is held to the oracle."""
    rng = np.random.default_rng(321)
    n, dim = 600_000, 32
    store = VecStore(dim)
    store.append_random(n, 99)
    rows = None
    for nq, k in ((3, 10), (40, 100), (256, 30)):
        queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
        mask = rng.random(n) < 0.3
        plans = {
            "plain": lambda p: store.query(queries, metric).take(k).per_query().with_path(p),
            "take_min": lambda p: store.query(queries, metric).take_min(k).per_query().with_path(p),
            "filter": lambda p: store.query(queries, metric).filter(0.2 if metric != Metric.Euclidean else 18.0, Cmp.Gt).take(k).per_query().with_path(p),
            "row_mask": lambda p: store.query(queries, metric).with_row_mask(mask).take(k).per_query().with_path(p),
        }
        for name, plan in plans.items():
            _, exact, _, _ = run(plan(Path.Exact))
            for spec in (1, 0):
                store.set_option("force_fallback", 0 if spec else 32)  # bit 32: conservative emission thresholds
                _, hits, _, stats = run(plan(Path.Mfma))
                assert stats["path_used"] == 2, (name, spec)
                assert_bit_exact(hits, exact)
            store.set_option("force_fallback", 0)
            if name == "plain" and nq == 3:
                rows = oracle.rand_rows(0, n, dim, 99) if rows is None else rows
                for q in range(nq):
                    rq = store.query(queries[q], metric).take(k).resolve()
                    ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
                    got = exact[exact["query"] == q]
                    assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))


def test_speculative_gate_on_sorted_corpora(oracle):
    """The gate's sample is the rows seen so far, so a corpus ORDERED by similarity is its worst case.  Best matches first: the
    gate lands among them, nothing later passes it, finalize sees fewer rows above the gate than it needs and reports the
    queries as failed by their gate (ott_stats.gate_failed); the next cascade level answers them with conservative gates and
    the store backs off (the next batch does not speculate).  Best matches last: the gate is loose, the last round lists
    them all.  Results equal the oracle's bit for bit in both orders."""
    rng = np.random.default_rng(77)
    n, dim, k, nq = 300_000, 48, 10, 24
    centre = rng.normal(0, 1, dim).astype(np.float32)
    near = (centre + rng.normal(0, 1, (8192, dim)) * rng.uniform(0.15, 6.0, (8192, 1))).astype(np.float32)  # cosines ~0.16 .. 0.99, about 1e-4 apart at the top
    far = rng.normal(0, 1, (n - 8192, dim)).astype(np.float32)
    queries = (centre + rng.normal(0, 0.05, (nq, dim))).astype(np.float32)
    for order, hi_fmt in (("best_first", 0), ("best_last", 0), ("best_first", 1), ("best_last", 1)):
        rows = np.concatenate([near, far] if order == "best_first" else [far, near])
        store = VecStore(dim)
        store.set_option("hi_fmt", hi_fmt)  # 0: bf16 hi plane (bound ~3e-3: the tight gate costs the certification); 1: IEEE half
        store.add_vectors(rows)
        plan = store.query(queries, Metric.Cosine).take(k).per_query().with_path(Path.Mfma)
        rq, hits, _, stats = run(plan)
        assert stats["path_used"] == 2
        for q in range(nq):
            rq1 = store.query(queries[q], Metric.Cosine).take(k).resolve()
            ref = oracle_collect(oracle, rq1, rows, oracle.TIES_CANONICAL)
            got = hits[hits["query"] == q]
            assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32)), (order, q)
        assert stats["retries"] == 0                                       # never the exact path
        if order == "best_first" and hi_fmt == 0:
            assert stats["gate_failed"] > 0                                 # answered by the next level
            _, hits2, _, stats2 = run(plan)                                 # backing off: no speculation, nothing fails
            assert stats2["gate_failed"] == 0 and stats2["refined"] == 0
            assert_bit_exact(hits2, hits)
        elif order == "best_first":
            # the half plane's bound (~4e-4) is tight enough to certify most of these queries even behind a gate that turned out
            # too tight; whichever fail are reported and answered by the next level, and the store then backs off
            _, hits2, _, stats2 = run(plan)
            assert stats2["gate_failed"] == 0 or stats["gate_failed"] == 0
            assert_bit_exact(hits2, hits)
        else:
            assert stats["gate_failed"] == 0
        store.close()


def test_a_violated_error_bound_is_noticed_and_answered_exactly(oracle):
    """The cascade certifies a query's top-k against a bound eps on |approximate - exact| that rests on a model of the matrix
    unit's accumulation.  The bound is CHECKED at run time: every re-scored candidate measures the difference, and a ratio
    above 1 voids the certification (ott_stats.bound_violations) — the query goes to the next level and finally to the
    exact-order kernel (src/vec_compute.rs:9-54).  With the bound deliberately shrunk (test option eps_scale_ppm) queries
    fall through and the result is still the oracle's, bit for bit; at scale 1 nothing is ever violated."""
    n, dim, nq, k = 200_000, 96, 48, 20
    rows = oracle.rand_rows(0, n, dim, 5)
    queries = oracle.rand_rows(0, nq, dim, 6)
    for hi_fmt in (1, 0):
        store = VecStore(dim)
        store.set_option("hi_fmt", hi_fmt)
        store.append_random(n, 5)
        for metric in (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
            plan = lambda: store.query(queries, metric).take(k).per_query().with_path(Path.Mfma)
            store.set_option("eps_scale_ppm", 1_000_000)
            _, good, _, st = run(plan())
            assert st["path_used"] == 2 and st["bound_violations"] == 0 and 0.0 < st["err_ratio_max"] <= 1.0, st
            seen = 0
            for ppm in (20_000, 1_000, 10):  # 1/50, 1/1000, 1/100000 of the bound
                store.set_option("eps_scale_ppm", ppm)
                _, hits, _, st2 = run(plan())
                assert st2["path_used"] == 2
                assert_bit_exact(hits, good)
                seen += st2["bound_violations"]
                if ppm == 10:
                    # a bound 100 000 times too small cannot hold for any query: every one is noticed, none is returned as
                    # certified — they all end on the exact-order kernel
                    assert st2["bound_violations"] >= nq and st2["retries"] == nq, st2
            assert seen > 0
            store.set_option("eps_scale_ppm", 1_000_000)
            for q in (0, 17, nq - 1):
                rq = store.query(queries[q], metric).take(k).resolve()
                ref = oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL)
                got = good[good["query"] == q]
                assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
        store.close()


def test_hi_plane_is_built_in_the_background_after_appends(oracle):
    """The 16-bit hi plane off the first batch's critical path (option hi_prebuild): stores of 262144 rows and more get it built
    by a background thread right after the appends — no query, no ott_store_prepare_batch — smaller stores and hi_prebuild = 0
    keep the lazy build inside the first batch.  Results are the same either way."""
    import time

    def wait_ready(store, seconds):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            if store.batch_ready():
                return True
            time.sleep(0.005)
        return store.batch_ready()

    dim, n = 64, 300_000
    queries = oracle.rand_rows(0, 24, dim, 9)
    big = VecStore(dim)
    big.append_random(n, 3)
    assert wait_ready(big, 10.0)                       # built without anybody asking
    big.append_random(50_000, 3)                       # ... and extended after further appends
    assert wait_ready(big, 10.0)
    lazy = VecStore(dim)
    lazy.set_option("hi_prebuild", 0)
    lazy.append_random(n + 50_000, 3)
    time.sleep(0.3)
    assert not lazy.batch_ready()
    a, _ = big.query(queries, Metric.Cosine).take(10).with_path(Path.Mfma).collect_arrays()
    assert big.last_stats["path_used"] == 2
    b, _ = lazy.query(queries, Metric.Cosine).take(10).with_path(Path.Mfma).collect_arrays()
    assert lazy.batch_ready()                           # the first batch built it
    e, _ = lazy.query(queries, Metric.Cosine).take(10).with_path(Path.Exact).collect_arrays()
    assert_bit_exact(a, e)
    assert_bit_exact(b, e)
    small = VecStore(dim)
    small.append_random(20_000, 3)
    time.sleep(0.3)
    assert not small.batch_ready()                      # automatic: small stores do not get a second copy nobody asked for
    forced = VecStore(dim)
    forced.set_option("hi_prebuild", 1)
    forced.append_random(20_000, 3)
    assert wait_ready(forced, 10.0)
    multi = VecStore(dim, devices=[0, 0])               # every shard of a multi-GPU store builds its own
    multi.reserve(2 * n)
    multi.append_random(2 * n, 3)
    assert wait_ready(multi, 10.0)
    for s in (big, lazy, small, forced, multi):
        s.close()


def test_auto_single_query_uses_a_resident_hi_plane(oracle):
    """AUTO sends one query down the exact-order kernel — unless the plane the cascade starts with (the int8 plane since round 5,
    the 16-bit hi plane before) is already resident (prepare_batch, the background build or an earlier batch made it) and the
    store is large enough for a quarter of the bytes to pay for the cascade's fixed ~0.25 ms: then the cascade answers it, same
    bits (profiles/round5/auto_choice.md: the two paths cross at ~500k x 768 rows; 10M rows: 1.45 ms against 4.7)."""
    n, dim = 1_200_000, 768  # 3.7 GB of rows
    store = VecStore(dim)
    store.set_option("hi_prebuild", 0)  # (the background build after appends would make the plane resident by itself: its own test)
    store.append_random(n, 41)
    q = oracle.rand_rows(0, 1, dim, 42)[0]
    plan = lambda: store.query(q, Metric.Cosine).take(10)
    _, first, _, stats = run(plan())
    assert stats["path_used"] == 1
    store.prepare_batch()
    _, second, _, stats = run(plan())
    assert stats["path_used"] == 2 and stats["retries"] == 0
    assert_bit_exact(second, first)
    store.add_vectors(oracle.rand_rows(0, 10, dim, 43))  # the plane no longer covers every row: back to the exact-order kernel
    _, _, _, stats = run(plan())
    assert stats["path_used"] == 1
    small = VecStore(128)  # a store where the exact kernel is the faster one stays there, plane or not
    small.append_random(100_000, 7)
    small.prepare_batch()
    _, _, _, stats = run(small.query(oracle.rand_rows(0, 1, 128, 8)[0], Metric.Cosine).take(10))
    assert stats["path_used"] == 1


def test_hi_pass_serves_larger_k_on_the_half_plane(oracle):
    """With the hi plane in IEEE half the first candidate pass also serves 228 < k <= 363 (its bound is tight enough for
    k + k / 3 + 28 <= 512 re-scored candidates); with bf16 those batches start at the split pass.  Same bits either way."""
    n, dim, nq = 200_000, 64, 16
    store = VecStore(dim)
    store.append_random(n, 31)
    rows = oracle.rand_rows(0, n, dim, 31)
    queries = oracle.rand_rows(0, nq, dim, 32)
    for k in (229, 300, 363, 364):
        hits, counts = store.query(queries, Metric.Cosine).take(k).per_query().with_path(Path.Mfma).collect_arrays()
        st = dict(store.last_stats)
        assert st["path_used"] == 2 and counts == [k] * nq and st["retries"] == 0, (k, st)
        for qi in (0, nq - 1):
            ref = oracle.vec_query(rows, queries[qi], oracle.METRIC_COSINE, oracle.TAKE_MAX, k, ties=oracle.TIES_CANONICAL, fast=True)
            g = hits[qi * k:(qi + 1) * k]
            assert np.array_equal(g["index"], ref["index"]) and np.array_equal(g["score"].view(np.uint32), ref["score"].view(np.uint32)), (k, qi)
    store.close()


def test_half_plane_with_large_norm_dot_and_l2_queries(oracle):
    """Rows and queries with norms in the tens of thousands: for dot / squared L2 the half plane's query operands (raw queries x
    the reciprocal of the plane's power-of-two factor) leave half's range, the hi pass is skipped and the split pass answers;
    cosine (unit-length operands) still certifies in the hi pass.  Same bits as the exact path throughout."""
    rng = np.random.default_rng(321)
    n, dim, nq = 50_000, 64, 16
    for scale in (1.0, 300.0, 40_000.0):
        rows = (rng.normal(0, 1, (n, dim)) * scale).astype(np.float32)
        queries = (rng.normal(0, 1, (nq, dim)) * scale).astype(np.float32)
        store = VecStore(dim)
        store.add_vectors(rows)
        for metric in (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
            a, ca = store.query(queries, metric).take(10).per_query().with_path(Path.Mfma).collect_arrays()
            st = dict(store.last_stats)
            b, cb = store.query(queries, metric).take(10).per_query().with_path(Path.Exact).collect_arrays()
            assert st["path_used"] == 2 and ca == cb, (scale, metric, st)
            assert np.array_equal(a["index"], b["index"]) and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)), (scale, metric)
            if metric == Metric.Cosine or scale <= 300.0:
                assert st["refined"] <= 1 and st["retries"] == 0, (scale, metric, st)  # certified by the hi pass (bar a near-tie)
        store.close()


def test_mfma_accumulation_error_probe():
    """tests/hip/mfma_accum_probe: the three MFMA instructions the candidate passes use, fed operands that are EXACT in their
    input format, accumulated over K = 96 / 768 / 3072 the way the kernels do, every one of the 32 x 32 outputs compared with the
    f64 sum.  The certification prices the matrix unit's summation at (1.25 .. 3.75) x K x 2^-24 of sum |a_i b_i| —
    an accumulation model AMD does not document; the probe fails if any output is off by more than K x 2^-24 of that (observed:
    a few units at most, same-sign and mixed-sign operands alike)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "tests", "hip"), "-s"])
    r = subprocess.run([os.path.join(root, "tests", "hip", "mfma_accum_probe")], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "ALL WITHIN BOUND" in r.stdout and r.stdout.count("max |mfma - exact|") == 27


def test_single_query_int8_sweep(oracle):
    """Round 5: ONE query at the cascade's int8 level is a streaming sweep — exact_kernel<..., I8> over the int8 plane (v_dot4, a
    lane per row), the 128 best APPROXIMATE scores in the wave lists, merge, then the exact re-score and certification of
    finalize_kernel (run_i8_single) — instead of five rounds on the matrix cores.  Whatever it certifies must be the oracle's
    bits; what it cannot (near-duplicates: more than ~118 rows within its bound of the k-th score) falls to the next levels:
    shapes with dims that are not multiples of 128, cosine and dot, take_min, score filters, row masks, chunk masks of one and of
    many runs, rows outside the error model (zero, huge, non-finite), k up to the sweep's 24."""
    rng = np.random.default_rng(404)
    for n, dim in ((70_000, 96), (200_000, 200), (50_001, 768), (30_000, 1030)):
        rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        rows[5] = 0.0                      # a zero row (cosine 0)
        rows[77, :] = 1e19                 # a huge row: outside every pass's error model
        rows[78, 3] = np.inf
        rows[1234] *= 1e-3
        rows[4321, 0] = 500.0              # one huge element among small ones: the int8 row measures a large loss (bit 2)
        store = VecStore(dim)
        store.set_chunk_size(1000)
        store.add_vectors(rows)
        store.prepare_batch()
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        mask = rng.random(n) < 0.7
        n_chunks = (n + 999) // 1000
        cm_one = np.zeros(n_chunks, bool)
        cm_one[3:n_chunks - 2] = True
        cm_many = (np.arange(n_chunks) % 2) == 1
        for metric, take in ((Metric.Cosine, 1), (Metric.DotProduct, 1), (Metric.DotProduct, 0)):
            for k in (1, 10, 24):
                for variant in ("plain", "gt", "lt", "row_mask", "chunk_one", "chunk_many"):
                    plan = store.query(q, metric)
                    kw = {}
                    okw = {}
                    if variant == "gt":
                        plan = plan.filter(0.01, Cmp.Gt)
                        okw = dict(filter_cmp=int(Cmp.Gt), filter_thr=0.01)
                    if variant == "lt":
                        plan = plan.filter(0.05, Cmp.Lt)
                        okw = dict(filter_cmp=int(Cmp.Lt), filter_thr=0.05)
                    if variant == "row_mask":
                        plan = plan.with_row_mask(mask)
                        okw = dict(row_mask=mask)
                    plan = (plan.take(k) if take else plan.take_min(k)).with_path(Path.Mfma)
                    rq = plan.resolve()
                    cm = cm_one if variant == "chunk_one" else cm_many if variant == "chunk_many" else None
                    hits, _, st = store._run(rq, chunk_mask=cm)
                    assert st["path_used"] == 2, (n, dim, metric, k, variant)
                    r2, inv2 = rows, None
                    if cm is not None:  # the oracle scores the surviving chunks' rows
                        keep = np.repeat(cm, 1000)[:n]
                        okw = dict(row_mask=keep)
                    ref = oracle.vec_query(rows, q, int(metric), take, k, okw.get("filter_cmp", 0), okw.get("filter_thr", 0.0),
                                           row_mask=okw.get("row_mask"), ties=oracle.TIES_CANONICAL)
                    where = (n, dim, metric, take, k, variant)
                    assert np.array_equal(hits["index"], ref["index"]), (where, hits[:5], ref[:5])
                    assert np.array_equal(hits["score"].view(np.uint32), ref["score"].view(np.uint32)), where
        store.close()
    # near-duplicates: hundreds of rows within the bound of the k-th score — the sweep cannot certify, the later levels answer
    base = rng.uniform(-1, 1, 256).astype(np.float32)
    rows = (base[None, :] + rng.normal(0, 1e-4, (20_000, 256))).astype(np.float32)
    store = VecStore(256)
    store.add_vectors(rows)
    store.prepare_batch()
    hits, _ = store.query(base, Metric.Cosine).take(10).with_path(Path.Mfma).collect_arrays()
    ref = oracle.vec_query(rows, base, oracle.METRIC_COSINE, oracle.TAKE_MAX, 10, ties=oracle.TIES_CANONICAL)
    assert np.array_equal(hits["index"], ref["index"]) and np.array_equal(hits["score"].view(np.uint32), ref["score"].view(np.uint32))
    assert store.last_stats["i8_refined"] == 1  # (left open by the int8 sweep, answered further down the cascade)
    store.close()


def test_int8_level_takes_squared_l2(oracle, monkeypatch):
    """Round 5: squared L2 at the cascade's int8 level — score ~ ||q||^2 + ||v||^2 - 2 (acc x s_v x s_Q), the norms exact, the dot
    product's quantisation loss measured as for cosine / dot and priced twice in the bound.  On uniform rows the level certifies
    every query by itself (nothing refined, nothing re-run), at every tile width, for nearest and farthest, with a distance
    filter and a row mask; the result is the oracle's bit for bit."""
    for k in ("OTT_MFMA_F32", "OTT_NO_BATCH_IMAGE", "OTT_NO_HI_PASS", "OTT_FORCE_FALLBACK", "OTT_HI_FMT"):
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(505)
    n, dim = 120_000, 264
    rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    rows[11] = 0.0
    store = VecStore(dim)
    store.set_chunk_size(1000)
    store.add_vectors(rows)
    mask = rng.random(n) < 0.6
    for nq in (1, 5, 20, 40, 100, 200, 300):
        queries = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
        for variant in ("nearest", "farthest", "within", "row_mask"):
            plan = store.query(queries, Metric.Euclidean)
            if variant == "within":
                plan = plan.filter(float(dim) * 0.62, Cmp.Lt)
            if variant == "row_mask":
                plan = plan.with_row_mask(mask)
            plan = (plan.take(12) if variant == "farthest" else plan.take_min(12)).with_path(Path.Mfma)
            rq, hits, _, stats = run(plan)
            assert stats["path_used"] == 2
            assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
            assert stats["i8_refined"] == 0 and stats["refined"] == 0 and stats["retries"] == 0, (nq, variant, stats)
            assert stats["bound_violations"] == 0 and 0.0 < stats["err_ratio_max"] < 0.5, (nq, variant, stats)
    # one row with a huge element: the bound of dot and squared L2 scales with the store's largest norm, so the level may leave
    # queries to the next one — the result is still the oracle's
    rows[4321, 0] = 500.0
    store.write_rows(4321, rows[4321:4322])
    queries = rng.uniform(-1, 1, (33, dim)).astype(np.float32)
    rq, hits, _, stats = run(store.query(queries, Metric.Euclidean).take_min(12).with_path(Path.Mfma))
    assert_bit_exact(hits, oracle_collect(oracle, rq, rows, oracle.TIES_CANONICAL))
