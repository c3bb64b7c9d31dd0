"""GPU: the batch cascade on NON-uniform data at full size.  The first candidate pass (bf16 hi plane) certifies a query only
when few rows lie within ~3e-3 (relative) of its k-th score — true of i.i.d. rows, not of clustered corpora, where it is
the later levels (split-bf16 pass, its 4096-candidate form, the exact path) that answer.  Whatever level answers, the
result must be the exact-order kernel's, bit for bit (src/vec.rs:243-266 is one loop: there is no approximate mode to fall
back on).  Corpus: 10M x 768 clustered rows generated on the device (ott_store_append_clustered; counter-based, the
oracle rebuilds any row), queries = unseen members of the same clusters."""
import numpy as np
import pytest

from otters_amd import Metric, Path, VecStore

pytestmark = pytest.mark.gpu

N, DIM, SEED = 10_000_000, 768, 0xC1A57E


def _corpus(n_clusters, spread, aniso):
    store = VecStore(DIM)
    store.reserve(N)
    store.append_clustered(N, SEED, n_clusters, spread, aniso)
    return store


REGIMES = {
    # loose clusters, anisotropic spread: what sentence-embedding corpora look like; the hi pass mostly certifies
    "loose_4096_clusters": (4096, 0.45, 2.0),
    # ~100 near-duplicates per cluster (cosine ~0.999 inside a cluster): hundreds of rows within the hi pass's bound of the k-th score
    "near_duplicates_100k_clusters": (100_000, 0.04, 0.0),
}


@pytest.mark.parametrize("regime", list(REGIMES), ids=list(REGIMES))
def test_cascade_is_exact_on_clustered_10m(oracle, regime):
    n_clusters, spread, aniso = REGIMES[regime]
    store = _corpus(n_clusters, spread, aniso)
    try:
        rebuilt = oracle.clustered_rows(123_456, 3, DIM, SEED, n_clusters, spread, aniso)
        assert np.array_equal(store.rows(123_456, 3), rebuilt)  # the device generator and the oracle's agree bit for bit
        for nq, k in ((64, 10), (256, 100)):
            queries = oracle.clustered_rows(N + 1000, nq, DIM, SEED, n_clusters, spread, aniso)
            hits, counts = store.query(queries, Metric.Cosine).take(k).per_query().collect_arrays()
            st = dict(store.last_stats)
            assert st["path_used"] == 2 and counts == [k] * nq, st
            per = hits.reshape(nq, k)
            sample = sorted(set([0, 1, nq // 3, nq // 2, nq - 2, nq - 1]))
            ex, _ = store.query(queries[sample], Metric.Cosine).take(k).with_path(Path.Exact).per_query().collect_arrays()
            ex = ex.reshape(len(sample), k)
            for j, qi in enumerate(sample):
                assert np.array_equal(per[qi]["index"], ex[j]["index"]), (regime, nq, qi, st)
                assert np.array_equal(per[qi]["score"].view(np.uint32), ex[j]["score"].view(np.uint32)), (regime, nq, qi)
            for qi in (2, nq - 3):  # every hit re-derived by the oracle from the regenerated row
                sc = []
                for i in per[qi]["index"]:
                    row = oracle.clustered_rows(int(i), 1, DIM, SEED, n_clusters, spread, aniso)[0]
                    sc.append(oracle.cosine(queries[qi], row, oracle.inv_norms(queries[qi])[0], oracle.inv_norms(row)[0]))
                assert np.array_equal(per[qi]["score"].view(np.uint32), np.array(sc, np.float32).view(np.uint32))
                assert np.all(np.diff(per[qi]["score"]) <= 0)
            rng = np.random.default_rng(9)
            for start in rng.integers(0, N - 40_000, 2):  # sampled completeness
                blk = oracle.clustered_rows(int(start), 40_000, DIM, SEED, n_clusters, spread, aniso)
                for qi in (0, nq - 1):
                    s = oracle.vec_query(blk, queries[qi], oracle.METRIC_COSINE, oracle.TAKE_MAX, 1, fast=True)
                    assert s["score"][0] <= per[qi]["score"][-1] or (int(s["index"][0]) + int(start)) in set(per[qi]["index"].tolist())
            # merged (reference semantics) = canonical merge of the per-query lists
            m, _ = store.query(queries, Metric.Cosine).take(k).collect_arrays()
            flat = hits[np.lexsort((hits["query"], hits["index"], -hits["score"].astype(np.float64)))][:k]
            assert np.array_equal(m["index"], flat["index"]) and np.array_equal(m["query"], flat["query"])
            assert np.array_equal(m["score"].view(np.uint32), flat["score"].view(np.uint32))
        # squared L2 (take_min) on the same corpus: the cancellation-prone expansion meets near-duplicates
        q32 = oracle.clustered_rows(N + 5000, 32, DIM, SEED, n_clusters, spread, aniso)
        a, ca = store.query(q32, Metric.Euclidean).take(20).per_query().collect_arrays()
        assert store.last_stats["path_used"] == 2
        b, cb = store.query(q32[:8], Metric.Euclidean).take(20).with_path(Path.Exact).per_query().collect_arrays()
        assert ca == [20] * 32 and np.array_equal(a[:160]["index"], b["index"]) and np.array_equal(a[:160]["score"].view(np.uint32), b["score"].view(np.uint32))
    finally:
        store.close()
