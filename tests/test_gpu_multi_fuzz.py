"""GPU: randomised comparison of the in-process multi-GPU store (ott_store_create_multi, device lists of repeated ordinal 0)
with ONE single-GPU store holding the same rows: random sizes, dims, shard counts, chunk sizes, append patterns (with and
without a plan, in pieces, row by row), metrics, k on both sides of 512, result modes, filters, row and chunk masks, tie orders,
paths.  Every draw must return the same hits bit for bit.  OTT_MULTI_FUZZ_SEEDS=400 for a soak."""
import os

import numpy as np
import pytest

from otters_amd import Cmp, Metric, Path, VecStore

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("exchange_mode")]


def same_hits(a, b, where):
    assert a.shape == b.shape, (where, a.shape, b.shape)
    assert np.array_equal(a["index"], b["index"]), (where, a[:6], b[:6])
    assert np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)), (where, a[:6], b[:6])
    assert np.array_equal(a["query"], b["query"]), (where, a[:6], b[:6])


@pytest.mark.parametrize("seed", range(int(os.environ.get("OTT_MULTI_FUZZ_SEEDS", "24"))))
def test_multi_store_fuzz(seed):
    rng = np.random.default_rng(91_000 + seed)
    n = int(rng.choice([3, 40, 777, 5000, 20_011, 70_000]))
    dim = int(rng.choice([1, 3, 8, 17, 64, 100, 256]))
    shards = int(rng.choice([1, 2, 3, 5, 8]))
    cs = int(rng.choice([1, 7, 8, 64, 1000, 1024]))
    quant = bool(rng.random() < 0.35)  # quantised rows: exact score ties across shard boundaries
    rows = (rng.integers(-2, 3, (n, dim)) if quant else rng.uniform(-1, 1, (n, dim))).astype(np.float32)
    tie = str(rng.choice(["canonical", "canonical", "reference", "reference_chunked"]))
    # (tie order 2 takes any chunk size since round 5: 1-, 7- and 1000-row chunks included)
    one, many = VecStore(dim), VecStore(dim, devices=[0] * shards)
    for s in (one, many):
        s.set_chunk_size(cs)
        s.set_tie_order(tie)
    # the same appends on both: a plan or none, pieces of random sizes, some of them single rows
    if rng.random() < 0.5:
        for s in (one, many):
            s.reserve(n)
    at = 0
    while at < n:
        step = int(rng.choice([1, 1, 2, 50, 1000, 9000, n]))
        step = min(step, n - at)
        for s in (one, many):
            if step == 1:
                s.add_vector(rows[at])
            else:
                s.add_vectors(rows[at:at + step])
        at += step
        if rng.random() < 0.15 and at < n:  # a query in the middle of loading (the shards are balanced, then unbalanced again)
            q = rng.uniform(-1, 1, dim).astype(np.float32)
            a, _ = one.query(q, Metric.DotProduct).take(5).collect_arrays()
            b, _ = many.query(q, Metric.DotProduct).take(5).collect_arrays()
            same_hits(b, a, (seed, "mid-load", at))
    assert many.len() == one.len() == n and sum(c for _, _, c in many.shards()) == n
    for _ in range(6):
        nq = int(rng.choice([1, 1, 2, 5, 33]))
        q = (rng.integers(-2, 3, (nq, dim)) if quant else rng.uniform(-1, 1, (nq, dim))).astype(np.float32)
        q[np.all(q == 0, axis=1)] = 1.0
        metric = Metric(int(rng.integers(0, 3)))
        k = int(rng.choice([1, 3, 10, 100, 500, 513, 2000, n * nq + 3]))
        perq = bool(rng.random() < 0.4) and tie == "canonical"
        path = Path(int(rng.choice([0, 0, 1, 2]))) if dim >= 8 and k + 28 <= 512 else Path.Auto
        filt = None if rng.random() < 0.6 else (float(rng.uniform(-0.3, 0.3)), Cmp(int(rng.integers(1, 6))))
        mask = (rng.random(int(rng.integers(1, n + 1))) < 0.7) if rng.random() < 0.3 else None
        n_chunks = (n + cs - 1) // cs
        cmask = (rng.random(n_chunks) < 0.6) if rng.random() < 0.3 else None

        def plan(s):
            p = s.query(q if nq > 1 else q[0], metric)
            p = p.take_min(k) if rng_take == 0 else p.take_max(k)
            if perq:
                p = p.per_query()
            if filt:
                p = p.filter(*filt)
            if mask is not None:
                p = p.with_row_mask(mask)
            return p.with_path(path)
        rng_take = int(rng.integers(0, 2))
        ra = plan(one).resolve()
        a, ca, _ = one._run(ra, chunk_mask=cmask)
        b, cb, _ = many._run(plan(many).resolve(), chunk_mask=cmask)
        where = (seed, n, dim, shards, cs, tie, nq, int(metric), k, perq, int(path), filt, mask is not None, cmask is not None)
        same_hits(b, a, where)
        if perq:
            assert list(ca) == list(cb), where
    one.close()
    many.close()
