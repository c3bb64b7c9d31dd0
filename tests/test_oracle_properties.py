"""CPU: properties of the oracle itself (the checker must be trustworthy before it checks anything):
literal collector == canonical form modulo the tie freedom the reference leaves, agreement with a
float64 numpy model within the metric's tolerance, mask semantics, meta == vec equivalences, and
the ASan/UBSan self-test of the C code."""
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from helpers import same_modulo_ties

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_sanitizer_selftest():
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "SELFTEST OK" in p.stdout, p.stdout + p.stderr


@settings(max_examples=60, deadline=None)
@given(n=st.integers(0, 90), dim=st.sampled_from([1, 3, 8, 9, 17]), nq=st.integers(1, 4), k=st.integers(0, 40),
       metric=st.integers(0, 2), take=st.integers(0, 1), cmp=st.integers(0, 5), quant=st.booleans(), seed=st.integers(0, 10**6))
def test_literal_equals_canonical_modulo_ties(oracle, n, dim, nq, k, metric, take, cmp, quant, seed):
    rng = np.random.default_rng(seed)
    if quant:  # quantised values: many exact ties
        rows = rng.integers(-2, 3, (n, dim)).astype(np.float32)
        q = rng.integers(-2, 3, (nq, dim)).astype(np.float32)
    else:
        rows = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
        q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    thr = 0.0 if metric != 1 else float(dim) / 2
    mask = rng.random(max(n - 3, 0)) < 0.7 if seed % 2 else None
    a = oracle.vec_query(rows, q, metric, take, k, cmp, thr, row_mask=mask, ties=oracle.TIES_LITERAL)
    b = oracle.vec_query(rows, q, metric, take, k, cmp, thr, row_mask=mask, ties=oracle.TIES_CANONICAL)
    same_modulo_ties(a["index"], a["score"], b["index"], b["score"], a["query"], b["query"])
    # canonical order is a strict total order
    keys = list(zip((-b["score"] if take == 1 else b["score"]).tolist(), b["index"].tolist(), b["query"].tolist()))
    assert keys == sorted(keys) or any(s == 0 for s in b["score"])  # (+0.0 / -0.0 order by total_cmp, not by value)


def test_scores_match_float64_model(oracle):
    rng = np.random.default_rng(1)
    rows = rng.uniform(-1, 1, (500, 768)).astype(np.float32)
    q = rng.uniform(-1, 1, 768).astype(np.float32)
    r64, q64 = rows.astype(np.float64), q.astype(np.float64)
    for mode in (oracle.REDUCE_AVX, oracle.REDUCE_SEQ4):
        h = oracle.vec_query(rows, q, oracle.METRIC_COSINE, oracle.TAKE_MAX, 500, reduce_mode=mode)
        want = (r64 @ q64) / np.linalg.norm(r64, axis=1) / np.linalg.norm(q64)
        assert np.max(np.abs(h["score"] - want[h["index"]])) < 1e-5  # BASELINE tolerance on f32 scores
        h = oracle.vec_query(rows, q, oracle.METRIC_EUCLIDEAN, oracle.TAKE_MIN, 500, reduce_mode=mode)
        want = ((r64 - q64) ** 2).sum(axis=1)
        assert np.max(np.abs(h["score"] - want[h["index"]]) / want[h["index"]]) < 1e-5
    a = oracle.vec_query(rows, q, 0, 1, 500, reduce_mode=oracle.REDUCE_AVX)
    b = oracle.vec_query(rows, q, 0, 1, 500, reduce_mode=oracle.REDUCE_SEQ4)
    assert np.max(np.abs(np.sort(a["score"]) - np.sort(b["score"]))) < 1e-6  # the two reduce orders differ by ulps only


def test_meta_query_equals_vec_query_when_nothing_is_pruned(oracle):
    rng = np.random.default_rng(2)
    rows = rng.uniform(-1, 1, (1000, 24)).astype(np.float32)
    q = rng.uniform(-1, 1, (3, 24)).astype(np.float32)
    for cs in (1, 7, 64, 1000, 5000):
        for threads in (1, 3):
            m, stats = oracle.meta_query(rows, cs, q, 0, 1, 25, ties=oracle.TIES_CANONICAL, n_threads=threads)
            v = oracle.vec_query(rows, q, 0, 1, 25, ties=oracle.TIES_CANONICAL)
            assert np.array_equal(m, v)
            assert stats["vectors_compared"] == 3000 and stats["pruned_chunks"] == 0


def test_row_mask_shorter_than_store_keeps_the_rest(oracle):
    rows = np.eye(6, dtype=np.float32)
    h = oracle.vec_query(rows, rows[5], 2, 1, 6, row_mask=np.zeros(3, bool))  # src/vec.rs:234: missing bit => keep
    assert sorted(h["index"].tolist()) == [3, 4, 5]


def test_nan_scores_are_dropped_and_zero_norm_is_zero(oracle):
    rows = np.array([[np.nan, 1], [0, 0], [1, 0]], np.float32)
    h = oracle.vec_query(rows, [1.0, 0.0], 0, 1, 3)
    assert h["index"].tolist() == [2, 1] and h["score"].tolist() == [1.0, 0.0]  # src/vec_compute.rs:237, src/vec.rs:367
