"""CPU: the closed form behind the library's `tie_order = reference` (otters_amd/csrc/ott_ties.hip) against the oracle's
literal restatement of the TopKCollector (OTTO_TIES_LITERAL: strict-improvement inserts at the binary search's position,
src/vec_compute.rs:236-277) on heavily quantised data, where almost every cut runs through a group of equal scores.

The GPU side only ever supplies two things: the candidates ranked by (score, visit order) — one more than asked for — and, on
demand, the collector's fill phase (the first k passing pairs in visit order).  Here both are computed with numpy from the
oracle's scores, and `collector_result` below is the same rule the C++ applies; what is compared with the oracle is the
(row, query) SET of the result and the score sequence.  The GPU tests (tests/test_gpu_ties.py) hold the library itself to the
same oracle; this file pins the derivation, so a disagreement there is a kernel problem and one here a reasoning problem."""
import numpy as np
import pytest


def visit_rank(row, q, n, nq):
    """Position of (row, q) in the reference's scoring loop (src/vec.rs:222-303): blocks of eight rows, every query per
    block, lanes in order; then the remainder rows query by query."""
    full = (n // 8) * 8
    row = np.asarray(row, dtype=np.int64)
    q = np.asarray(q, dtype=np.int64)
    inblock = ((row >> 3) * nq + q) * 8 + (row & 7)
    rem = full * nq + q * (n - full) + (row - full)
    return np.where(row < full, inblock, rem)


def collector_result(L, k, fill):
    """L: list of (ordkey, visit, row, q) sorted by (ordkey, visit), up to k + 1 entries; fill: set of (row, q) of the fill
    phase.  Returns the collector's buffer (list of entries) — see the derivation in ott_ties.hip."""
    m = len(L)
    if m == 0 or k == 0:
        return []
    over = m > k
    kk = k if over else m
    ambiguous = over and L[kk][0] == L[kk - 1][0]
    in_f = lambda e: (e[2], e[3]) in fill  # noqa: E731
    g0 = kk
    while g0 > 0 and L[g0 - 1][0] == L[kk - 1][0]:
        g0 -= 1

    def arrange(run):  # [members in visit order without the anchor.., anchor]
        if len(run) < 2:
            return list(run)
        last_in = max((i for i, e in enumerate(run) if in_f(e)), default=None)
        a = last_in if last_in is not None else 0
        return run[:a] + run[a + 1:] + [run[a]]

    if ambiguous:
        c = kk - g0
        nxt = L[kk]
        inserted = any(b[1] > nxt[1] for b in L[:g0])  # a strictly better pair is visited after g_{c+1}
        if not inserted:
            cut = arrange(L[g0:kk])
        else:
            mf = 0
            while mf <= c and in_f(L[g0 + mf]):
                mf += 1
            if mf == c + 1:
                cut = L[g0:kk]
            else:
                a = mf - 1 if mf else 0
                cut = [L[g0 + i] for i in range(c + 1) if i != a]
    else:
        cut = arrange(L[g0:kk])
    out, i = [], 0
    while i < g0:
        j = i + 1
        while j < g0 and L[j][0] == L[i][0]:
            j += 1
        out += arrange(L[i:j])
        i = j
    return out + cut


def model_query(oracle, rows, queries, metric, take, k, cmp=0, thr=0.0, row_mask=None):
    n, nq = rows.shape[0], queries.shape[0]
    # every (row, query) score exactly as the oracle computes it: take = every pair, canonical order, then index back
    allh = oracle.vec_query(rows, queries, metric, take, n * nq, cmp, thr, row_mask=row_mask, ties=oracle.TIES_CANONICAL)
    bits = allh["score"].view(np.uint32).astype(np.int64)
    key = np.where(bits & 0x80000000, ~bits & 0xFFFFFFFF, bits | 0x80000000)
    ordkey = -key if take == oracle.TAKE_MAX else key  # ascending = better first
    vis = visit_rank(allh["index"].astype(np.int64), allh["query"].astype(np.int64), n, nq)
    order = np.lexsort((vis, ordkey))
    cand = [(int(ordkey[i]), int(vis[i]), int(allh["index"][i]), int(allh["query"][i])) for i in order[: k + 1]]
    by_visit = np.argsort(vis, kind="stable")[:k]
    fill = {(int(allh["index"][i]), int(allh["query"][i])) for i in by_visit}
    res = collector_result(cand, k, fill)
    score_of = {(int(allh["index"][i]), int(allh["query"][i])): allh["score"][i] for i in order[: k + 1]}
    return [(r, q) for _, _, r, q in res], np.array([score_of[(r, q)] for _, _, r, q in res], np.float32)


@pytest.mark.parametrize("seed", range(40))
def test_closed_form_equals_the_literal_collector(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([5, 8, 9, 23, 64, 100, 257]))
    dim = int(rng.choice([2, 3, 8, 11]))
    nq = int(rng.choice([1, 1, 2, 3, 5]))
    levels = int(rng.choice([2, 3, 4]))
    rows = rng.integers(-levels, levels + 1, (n, dim)).astype(np.float32)
    queries = rng.integers(-2, 3, (nq, dim)).astype(np.float32)
    metric = int(rng.choice([oracle.METRIC_COSINE, oracle.METRIC_EUCLIDEAN, oracle.METRIC_DOT]))
    take = oracle.TAKE_MIN if metric == oracle.METRIC_EUCLIDEAN and rng.random() < 0.8 else int(rng.choice([0, 1]))
    mask = (rng.random(n) < 0.8) if rng.random() < 0.4 else None
    cmp, thr = (0, 0.0)
    if rng.random() < 0.4:
        cmp, thr = int(rng.choice([1, 2, 3, 4])), float(rng.integers(-2, 3))
    for k in (1, 2, 3, 5, 8, 13, 20, 40, n * nq, n * nq + 3):
        lit = oracle.vec_query(rows, queries, metric, take, k, cmp, thr, row_mask=mask, ties=oracle.TIES_LITERAL)
        got, sc = model_query(oracle, rows, queries, metric, take, k, cmp, thr, mask)
        assert len(got) == lit.size, (seed, k, len(got), lit.size)
        assert sorted(got) == sorted(zip(lit["index"].tolist(), lit["query"].tolist())), (seed, k, n, nq, metric, take)
        assert np.array_equal(sc.view(np.uint32), lit["score"].view(np.uint32)), (seed, k)
        # and the buffer ORDER inside runs of equal scores (what tie_order = 2 merges by position)
        assert got == list(zip(lit["index"].tolist(), lit["query"].tolist())), (seed, k, n, nq)
