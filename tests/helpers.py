"""Shared test helpers: golden-case loading, expectation checks, tie-aware comparison."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

METRIC = {"cosine": 0, "euclidean": 1, "dot": 2}
CMP = {"lt": 1, "gt": 2, "lte": 3, "gte": 4, "eq": 5}


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def holds(score, op, thr):
    s, t = np.float32(score), np.float32(thr)
    return {"lt": s < t, "gt": s > t, "lte": s <= t, "gte": s >= t, "eq": s == t}[op]


def check_expect(idx, scores, exp):
    """idx / scores: result order.  exp: the expectation dict of a golden case."""
    idx = [int(i) for i in idx]
    scores = [float(s) for s in scores]
    tol = exp.get("tol", 1e-6)
    if "len" in exp:
        assert len(idx) == exp["len"], (idx, exp)
    if "len_le" in exp:
        assert len(idx) <= exp["len_le"]
    if exp.get("nonempty"):
        assert len(idx) > 0
    if "scores_by_index" in exp:
        for i, s in exp["scores_by_index"].items():
            got = [sc for ix, sc in zip(idx, scores) if ix == int(i)]
            assert got, f"index {i} missing from {idx}"
            assert abs(got[0] - s) < tol, (i, got[0], s)
    if "scores_by_index_loose" in exp:
        for i, s in exp["scores_by_index_loose"].items():
            got = [sc for ix, sc in zip(idx, scores) if ix == int(i)]
            assert got and abs(got[0] - s) < exp["tol_loose"]
    if "scores_in_order" in exp:
        assert len(scores) == len(exp["scores_in_order"])
        for a, b in zip(scores, exp["scores_in_order"]):
            assert abs(a - b) < tol, (scores, exp["scores_in_order"])
    if "score_bits_in_order" in exp:
        got = [hex(int(np.float32(s).view(np.uint32))) for s in scores]
        assert [g.lower() for g in got] == [e.lower() for e in exp["score_bits_in_order"]], got
    if "indices_in_order" in exp:
        assert idx == exp["indices_in_order"], idx
    if "indices_in_order_ties" in exp:  # groups of tied rows: order inside a group is unspecified
        o = 0
        for grp in exp["indices_in_order_ties"]:
            assert sorted(idx[o:o + len(grp)]) == sorted(grp), (idx, exp["indices_in_order_ties"])
            o += len(grp)
    if "index_set" in exp:
        assert sorted(idx) == sorted(exp["index_set"]), (idx, exp["index_set"])
    if exp.get("sorted") == "desc":
        assert all(scores[i - 1] >= scores[i] for i in range(1, len(scores)))
    if exp.get("sorted") == "asc":
        assert all(scores[i - 1] <= scores[i] for i in range(1, len(scores)))
    if "all_scores" in exp:
        op, thr = exp["all_scores"]
        assert all(holds(s, op, thr) for s in scores), (scores, exp["all_scores"])
    if "count_score" in exp:
        val, cnt = exp["count_score"]
        assert sum(1 for s in scores if abs(s - val) < tol) == cnt


def check_stats(stats, exp):
    for key, val in exp.items():
        if key.endswith("_ge"):
            assert stats[key[:-3]] >= val, (key, stats)
        elif key == "evaluated_le_total":
            assert stats["evaluated_chunks"] <= stats["total_chunks"]
        else:
            assert stats[key] == val, (key, stats)


def same_modulo_ties(a_idx, a_sc, b_idx, b_sc, a_q=None, b_q=None):
    """Two result lists are equivalent under the reference's contract when the score sequences
    are identical and, inside each run of equal scores that is fully contained in both lists,
    the (index, query) multisets agree.  The LAST run may be cut by k: there any tied
    candidate is acceptable, so only the scores are compared."""
    a_sc = np.asarray(a_sc, dtype=np.float32)
    b_sc = np.asarray(b_sc, dtype=np.float32)
    assert a_sc.shape == b_sc.shape, (a_sc.shape, b_sc.shape)
    assert np.array_equal(a_sc, b_sc), "score sequences differ"
    n = a_sc.size
    i = 0
    while i < n:
        j = i
        while j < n and a_sc[j] == a_sc[i]:
            j += 1
        if j < n:  # run not cut by k
            ka = sorted(zip(map(int, a_idx[i:j]), map(int, a_q[i:j]) if a_q is not None else [0] * (j - i)))
            kb = sorted(zip(map(int, b_idx[i:j]), map(int, b_q[i:j]) if b_q is not None else [0] * (j - i)))
            assert ka == kb, (i, j, ka, kb)
        i = j


# ---- golden-case drivers --------------------------------------------------------------------
def plan_from_case(case, store):
    """Builds the VecQueryPlan a golden vec case describes, on the given VecStore."""
    from otters_amd import Cmp, Metric
    metric = {"cosine": Metric.Cosine, "euclidean": Metric.Euclidean, "dot": Metric.DotProduct}[case["metric"]]
    plan = store.query(case["queries"], metric)
    if case.get("filter"):
        thr, op = case["filter"]
        plan = plan.filter(thr, {"lt": Cmp.Lt, "gt": Cmp.Gt, "lte": Cmp.Lte, "gte": Cmp.Gte, "eq": Cmp.Eq}[op])
    for kind, k in case.get("take", []):
        plan = getattr(plan, kind)(k)
    return plan


def oracle_collect(O, rq, rows, ties, reduce_mode=0):
    """Runs a ResolvedQuery on the CPU oracle (the checker)."""
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, rq.queries.shape[1])
    return O.vec_query(rows, rq.queries, rq.metric, rq.take, rq.k, rq.filter_cmp, rq.filter_thr, row_mask=rq.row_mask,
                       reduce_mode=reduce_mode, ties=ties)


# ---- MetaStore golden-case drivers ----------------------------------------------------------------
def expr_from_json(j):
    from otters_amd import col
    if j[0] == "cmp":
        return getattr(col(j[1]), j[2])(j[3])
    a, b = expr_from_json(j[1]), expr_from_json(j[2])
    return (a & b) if j[0] == "and" else (a | b)


def build_meta_case(case, host_only):
    from otters_amd import Column, DataType, MetaStore
    cols = []
    for c in case["columns"]:
        cols.append(Column(c["name"], DataType[c["dtype"]]).from_(c["values"]))
    return MetaStore.from_columns(cols).with_vectors(case["vectors"]).with_chunk_size(case["chunk_size"]).build(_host_only=host_only)


def meta_plan_from_case(case, meta):
    from otters_amd import Cmp, Metric
    metric = {"cosine": Metric.Cosine, "euclidean": Metric.Euclidean, "dot": Metric.DotProduct}[case["metric"]]
    q = case["queries"]
    plan = meta.query_batch(q, metric) if isinstance(q[0], list) else meta.query(q, metric)
    if case.get("meta_filter"):
        plan = plan.meta_filter(expr_from_json(case["meta_filter"]))
    if case.get("vec_filter"):
        thr, op = case["vec_filter"]
        plan = plan.vec_filter(thr, {"lt": Cmp.Lt, "gt": Cmp.Gt, "lte": Cmp.Lte, "gte": Cmp.Gte, "eq": Cmp.Eq}[op])
    if case.get("take") is not None:
        plan = plan.take(case["take"])
    return plan


# ---- exchange modes of the in-process multi-GPU store on a one-GPU box (tests/conftest.py: fixture `exchange_mode`) -------------
FAKE_RCCL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfake_rccl.so")


def multi_mode() -> str:
    """the mode the running test's stores are created in: "local", "remote" or "fake_rccl" (what the fixture put in the environment)"""
    if os.environ.get("OTT_MULTI_FAKE_DISTINCT") != "1":
        return "local"
    return "remote" if os.environ.get("OTT_MULTI_TRANSPORT") == "1" else "fake_rccl"
