"""Stand-alone check of ONE store over several DISTINCT GPUs of this process (ott_store_create_multi with different device
ordinals: peer copies / RCCL over xGMI, row moves between GPUs, the merge on the first GPU) against one single-GPU store with
the same rows — hits equal bit for bit (index, f32 score bits, query).

    python tests/multi_devices_check.py                      # every GPU of the machine, both transports
    python tests/multi_devices_check.py --devices 0,1,2,3 --transport rccl

tests/test_gpu_multi_devices.py runs it in a child process (a hang or a crash of a never-before-seen GPU topology must not take
the test session with it) when the machine has two GPUs or more; on a one-GPU box it runs it with the list 0,0 (same code, no
second device).  Prints one `OK ...` line per configuration and exits 0, or raises."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("OTT_MULTI_MIN_SHARD_ROWS", "0")  # every shard gets rows, however small the store (the default keeps small stores on one GPU)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from otters_amd import Cmp, Column, DataType, MetaStore, Metric, Path, VecStore, col  # noqa: E402
from otters_amd import _native as N  # noqa: E402

TRANSPORT = {"auto": 0, "peer": 1, "rccl": 2}


def same_hits(a, b, where):
    assert a.shape == b.shape, (where, a.shape, b.shape)
    assert np.array_equal(a["index"], b["index"]), (where, a[:8], b[:8])
    assert np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)), (where, a[:8], b[:8])
    assert np.array_equal(a["query"], b["query"]), (where, a[:8], b[:8])


def check(devs, transport):
    rng = np.random.default_rng(len(devs) * 7 + TRANSPORT[transport])
    G = len(devs)
    cases = 0

    # --- planned layout, synthetic rows generated on every GPU
    n, dim = 200_000, 96
    one, many = VecStore(dim, device=devs[0]), VecStore(dim, devices=devs)
    for s in (one, many):
        s.reserve(n)
        s.append_random(n, 21)
    many.set_option("multi_transport", TRANSPORT[transport])
    sh = many.shards()
    assert [d for d, _, _ in sh] == list(devs) and sum(c for _, _, c in sh) == n, sh
    assert np.array_equal(many.rows(n // G - 100, 200), one.rows(n // G - 100, 200))  # across the first boundary
    assert np.array_equal(many.inv_norms().view(np.uint32), one.inv_norms().view(np.uint32))
    q1 = rng.uniform(-1, 1, dim).astype(np.float32)
    q5 = rng.uniform(-1, 1, (5, dim)).astype(np.float32)
    q300 = rng.uniform(-1, 1, (300, dim)).astype(np.float32)
    for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
        for k in (1, 10, 100, 512, 513, 3000):
            for q in (q1, q5):
                a, _ = one.query(q, metric).take(k).collect_arrays()
                b, _ = many.query(q, metric).take(k).collect_arrays()
                same_hits(b, a, (metric, k, q.shape))
                cases += 1
        for k in (7, 600):
            a, ca = one.query(q5, metric).per_query().take(k).collect_arrays()
            b, cb = many.query(q5, metric).per_query().take(k).collect_arrays()
            same_hits(b, a, ("per query", metric, k))
            assert ca == cb
            cases += 1
    used = many.transport()
    if transport != "auto":
        assert used == transport, (used, transport)
    # score filter, host row mask shorter than the store
    a, _ = one.query(q5, Metric.Cosine).filter(0.05, Cmp.Gt).take_min(50).collect_arrays()
    b, _ = many.query(q5, Metric.Cosine).filter(0.05, Cmp.Gt).take_min(50).collect_arrays()
    same_hits(b, a, "filter")
    mask = rng.random(n - 4321) < 0.3
    for k in (10, 700):
        a, _ = one.query(q5, Metric.DotProduct).with_row_mask(mask).take(k).collect_arrays()
        b, _ = many.query(q5, Metric.DotProduct).with_row_mask(mask).take(k).collect_arrays()
        same_hits(b, a, ("row mask", k))
        cases += 1
    # the matrix-core cascade on every GPU, 300 queries (two query blocks), merged and per query
    a, _ = one.query(q300, Metric.Cosine).take(100).with_path(Path.Exact).collect_arrays()
    b, _ = many.query(q300, Metric.Cosine).take(100).with_path(Path.Mfma).collect_arrays()
    same_hits(b, a, "cascade")
    assert many.last_stats["path_used"] == 2 and many.last_stats["bound_violations"] == 0
    a, ca = one.query(q300, Metric.Euclidean).per_query().take(20).with_path(Path.Exact).collect_arrays()
    b, cb = many.query(q300, Metric.Euclidean).per_query().take(20).with_path(Path.Auto).collect_arrays()
    same_hits(b, a, "cascade per query")
    cases += 2
    # queries from several host threads at once (the reference's query is `&self`)
    import threading
    errs = []

    def worker(i):
        try:
            qq = np.random.default_rng(100 + i).uniform(-1, 1, (3, dim)).astype(np.float32)
            for _ in range(10):
                x, _ = one.query(qq, Metric.Cosine).take(20).collect_arrays()
                y, _ = many.query(qq, Metric.Cosine).take(20).collect_arrays()
                same_hits(y, x, ("thread", i))
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    cases += 1
    one.close()
    many.close()

    # --- no plan: host rows land in the first shard, the query moves them between the GPUs; write_rows across a boundary
    dim = 33
    rows = rng.uniform(-1, 1, (20_000, dim)).astype(np.float32)
    one, many = VecStore(dim, device=devs[0]), VecStore(dim, devices=devs)
    many.set_option("multi_transport", TRANSPORT[transport])
    for lo, hi in ((0, 9000), (9000, 9001), (9001, 20_000)):
        one.add_vectors(rows[lo:hi])
        many.add_vectors(rows[lo:hi])
        q = rng.uniform(-1, 1, (3, dim)).astype(np.float32)
        a, _ = one.query(q, Metric.Cosine).take(25).collect_arrays()
        b, _ = many.query(q, Metric.Cosine).take(25).collect_arrays()
        same_hits(b, a, ("after appends", hi))
        cases += 1
    cnt = [c for _, _, c in many.shards()]
    assert sum(cnt) == 20_000 and max(cnt) <= 20_000 // G + 1024 + 20_000 // (4 * G), cnt
    assert np.array_equal(many.rows(), rows)
    if G > 1 and many.shards()[1][2] > 0:
        first = many.shards()[1][1] - 3
        new = rng.uniform(-1, 1, (7, dim)).astype(np.float32)
        one.write_rows(first, new)
        many.write_rows(first, new)
        b, _ = many.query(new[3], Metric.Cosine).take(5).collect_arrays()
        a, _ = one.query(new[3], Metric.Cosine).take(5).collect_arrays()
        same_hits(b, a, "write_rows")
        assert int(b["index"][0]) == first + 3
    # device rows of the first GPU appended to a store whose later shards live elsewhere (ott_store_append_device)
    one.close()
    many.close()

    # --- the reference's tie outcomes across GPUs: quantised rows, cuts through groups of equal scores
    n, dim, cs = 20_011, 8, 256
    rows = rng.integers(-2, 3, (n, dim)).astype(np.float32)
    queries = rng.integers(-2, 3, (2, dim)).astype(np.float32)
    queries[np.all(queries == 0, axis=1)] = 1.0
    one, many = VecStore(dim, device=devs[0]), VecStore(dim, devices=devs)
    many.set_option("multi_transport", TRANSPORT[transport])
    for s in (one, many):
        s.set_chunk_size(cs)
        s.set_tie_order("reference")
        s.add_vectors(rows)
    for metric in (Metric.DotProduct, Metric.Euclidean, Metric.Cosine):
        for k in (1, 20, 150, 512, 700):
            a, _ = one.query(queries, metric).take(k).collect_arrays()
            b, _ = many.query(queries, metric).take(k).collect_arrays()
            same_hits(b, a, ("tie_order 1", metric, k))
            cases += 1
    for s in (one, many):
        s.set_tie_order("reference_chunked")
    cm = (np.arange((n + cs - 1) // cs) % 3) != 1
    for metric in (Metric.DotProduct, Metric.Cosine):
        for k in (4, 30, 100):
            rq = many.query(queries, metric).take(k).resolve()
            a, _, _ = one._run(rq, chunk_mask=cm)
            b, _, _ = many._run(rq, chunk_mask=cm)
            same_hits(b, a, ("tie_order 2", metric, k))
            cases += 1
    one.close()
    many.close()

    # --- MetaStore: columns and zone statistics per GPU, device row masks stitched together, zonemap-pruned queries
    n, dim, cs = 30_011, 48, 256
    vec = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    chunk = np.arange(n) // cs

    def cols():
        r = np.random.default_rng(77)
        return [Column.from_numpy("price", DataType.Float64, (chunk % 5) * 20.0 + r.uniform(0, 25, n), r.random(n) < 0.05),
                Column.from_numpy("version", DataType.Int32, (chunk % 3) + r.integers(0, 2, n), r.random(n) < 0.05),
                Column.from_numpy("grade", DataType.String, np.array(["A", "B", "C", "D"])[(chunk + r.integers(0, 2, n)) % 4], r.random(n) < 0.03)]
    m_one = MetaStore.from_columns(cols(), devices=[devs[0]]).with_vectors(vec).with_chunk_size(cs).build()
    m_many = MetaStore.from_columns(cols(), devices=list(devs)).with_vectors(vec).with_chunk_size(cs).build()
    for name in ("price", "version"):
        za, zb = m_one._zones[name], m_many._zones[name]
        assert np.array_equal(za.min, zb.min) and np.array_equal(za.max, zb.max) and np.array_equal(za.non_null, zb.non_null), name
    q = rng.uniform(-1, 1, (4, dim)).astype(np.float32)
    for f in (lambda: col("price").lt(50.0) & col("version").gte(2), lambda: col("grade").eq("A") | col("grade").eq("B")):
        for k in (10, 900):
            ra = m_one.query_batch(q, Metric.Cosine).meta_filter(f()).vec_filter(-0.2, Cmp.Gt).take(k).collect()
            rb = m_many.query_batch(q, Metric.Cosine).meta_filter(f()).vec_filter(-0.2, Cmp.Gt).take(k).collect()
            assert ra.indices == rb.indices and len(rb.indices) > 0
            assert np.array_equal(np.array(ra.scores, np.float32).view(np.uint32), np.array(rb.scores, np.float32).view(np.uint32))
            sa, sb = m_one.last_query_stats(), m_many.last_query_stats()
            assert (sa.pruned_chunks, sa.evaluated_chunks, sa.vectors_compared) == (sb.pruned_chunks, sb.evaluated_chunks, sb.vectors_compared)
            cases += 1
    print(f"OK devices {list(devs)} transport {transport} -> {used}: {cases} comparisons equal", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="", help="comma-separated device ordinals (default: every GPU of the machine)")
    ap.add_argument("--transport", default="", help="auto | peer | rccl (default: peer and, for distinct devices, rccl and auto)")
    args = ap.parse_args()
    N.lib()
    if args.devices:
        devs = [int(x) for x in args.devices.split(",")]
    else:
        import ctypes as C
        cnt = C.c_int(0)
        N.check(N.lib().ott_device_count(C.byref(cnt)))
        devs = list(range(cnt.value))
    distinct = len(set(devs)) == len(devs)
    transports = [args.transport] if args.transport else (["peer", "rccl", "auto"] if distinct and len(devs) > 1 else ["peer"])
    if "rccl" in transports or "auto" in transports:
        N.preload_torch_rccl()
    lists = [devs] if args.devices or len(devs) <= 2 else [devs[:2], devs]
    for d in lists:
        for t in transports:
            check(d, t)
    print("ALL OK", flush=True)


if __name__ == "__main__":
    main()
