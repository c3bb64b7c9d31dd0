"""GPU: the split-bf16 candidate pass against data built to stress its error bound — heavy-tailed elements (each element
scaled by exp(N(0, 3)): a few coordinates carry the norm), massive cancellation (rows = ±the query plus small noise), values
near bf16 rounding boundaries, subnormal-scale rows next to huge ones.  The batch path must agree with the exact path bit
for bit (the certification either holds or sends the query to the exact path; it must never return a wrong list)."""
import numpy as np
import pytest

from otters_amd import Metric, Path, VecStore

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["cascade", "split_pass_only"])
def candidate_passes(request, monkeypatch):
    """every test here runs twice: with the default cascade (bf16 hi pass first) and with the split pass alone"""
    monkeypatch.delenv("OTT_NO_HI_PASS", raising=False)
    if request.param == "split_pass_only":
        monkeypatch.setenv("OTT_NO_HI_PASS", "1")


def _agree(store, queries, k):
    for metric in (Metric.Cosine, Metric.DotProduct, Metric.Euclidean):
        a, ca = store.query(queries, metric).per_query().take(k).with_path(Path.Mfma).collect_arrays()
        assert store.last_stats["path_used"] == 2
        b, cb = store.query(queries, metric).per_query().take(k).with_path(Path.Exact).collect_arrays()
        assert ca == cb
        assert np.array_equal(a["index"], b["index"]), metric
        assert np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)), metric


@pytest.mark.parametrize("seed", range(6))
def test_split_bf16_error_bound_holds_on_adversarial_data(seed):
    rng = np.random.default_rng(500 + seed)
    n, dim, nq = 40_000, [64, 200, 768][seed % 3], [20, 70, 260][seed // 2]
    kind = seed % 3
    queries = rng.normal(0, 1, (nq, dim)).astype(np.float32)
    if kind == 0:      # heavy tails: a handful of coordinates dominate every norm
        rows = (rng.normal(0, 1, (n, dim)) * np.exp(rng.normal(0, 3, (n, dim)))).astype(np.float32)
        queries = (queries * np.exp(rng.normal(0, 3, (nq, dim)))).astype(np.float32)
    elif kind == 1:    # cancellation: rows are +-queries plus noise at 1e-3, so scores cluster at +-1 and 0
        base = queries[rng.integers(0, nq, n)] * rng.choice([-1.0, 1.0], (n, 1))
        rows = (base + rng.normal(0, 1e-3, (n, dim))).astype(np.float32)
    else:              # bf16 rounding boundaries and a wide range of row scales
        rows = rng.integers(-512, 513, (n, dim)).astype(np.float32) / 256.0 + np.float32(2.0 ** -9)
        rows *= np.exp2(rng.integers(-60, 60, (n, 1))).astype(np.float32)
    store = VecStore(dim)
    store.add_vectors(rows)
    _agree(store, queries, 10)
    _agree(store, queries[: max(nq // 3, 5)], 100)


def test_tiny_and_huge_rows_are_always_rescored_exactly():
    """Rows (and queries) whose norm underflows, is below 1e-18 or above 1e18 are outside the error model of the candidate
    pass: flagged at append, they are re-scored exactly for every query, so the batch path still equals the exact path —
    including a store made ONLY of such rows."""
    rng = np.random.default_rng(9)
    dim, nq = 96, 12
    queries = rng.normal(0, 1, (nq, dim)).astype(np.float32)
    normal = rng.normal(0, 1, (3000, dim)).astype(np.float32)
    tiny = (rng.normal(0, 1, (500, dim)) * 1e-30).astype(np.float32)      # squares underflow: norm computes to 0
    small = (rng.normal(0, 1, (500, dim)) * 1e-20).astype(np.float32)     # norm ~1e-19
    huge = (rng.normal(0, 1, (50, dim)) * 1e19).astype(np.float32)        # norm ~1e20
    for rows in (np.concatenate([normal, tiny, small, huge]), np.concatenate([tiny, small]), tiny):
        rows = rows[rng.permutation(rows.shape[0])]
        store = VecStore(dim)
        store.add_vectors(rows)
        _agree(store, queries, 10)
    q2 = np.concatenate([queries[:4], queries[4:8] * np.float32(1e-25), queries[8:] * np.float32(1e19)])
    store = VecStore(dim)
    store.add_vectors(np.concatenate([normal, small]))
    _agree(store, q2, 10)
