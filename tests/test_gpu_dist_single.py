"""GPU: the sharded path (ott_query_device -> RCCL all_gather -> ott_merge_hits_device) on a
1-rank process group: same kernels and the same exchange code as N ranks, checked against the
plain single-store query and the oracle.  (N > 1 logic: tests/test_dist_cpu.py with gloo.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_store_single_rank(oracle):
    import torch
    import torch.distributed as dist
    from otters_amd import Cmp, Metric, VecStore
    from otters_amd.dist import ShardedVecStore
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, dim, base = 30000, 64, 5_000_000
        store = VecStore(dim)
        store.set_base_offset(base)
        store.append_random(n, seed=3)
        rows = oracle.rand_rows(base, n, dim, 3)
        sh = ShardedVecStore(store, dist)
        q = np.random.default_rng(2).uniform(-1, 1, (2, dim)).astype(np.float32)
        for metric, k in ((Metric.Cosine, 10), (Metric.Euclidean, 100), (Metric.DotProduct, 130), (Metric.Cosine, 300)):  # 300: 8 list entries per lane
            got = sh.query(q, metric).take(k).collect()
            want = store.query(q, metric).take(k).collect()
            assert got == want
            ref = oracle.vec_query(rows, q, int(metric), 0 if metric == Metric.Euclidean else 1, k, ties=oracle.TIES_CANONICAL)
            assert [r.index - base for r in got] == [int(i) for i in ref["index"]]
            assert np.array_equal(np.array([r.score for r in got], np.float32).view(np.uint32), ref["score"].view(np.uint32))
        got = sh.query(q[0], Metric.Cosine).filter(0.9, Cmp.Gt).take(5).collect()
        assert got == []
        # PER_QUERY through the same exchange: [nq, k] blocks, grouped device merge
        qs = np.random.default_rng(4).uniform(-1, 1, (7, dim)).astype(np.float32)
        for k in (10, 100):
            got = sh.query(qs, Metric.Cosine).per_query().take(k).collect()
            want = store.query(qs, Metric.Cosine).per_query().take(k).collect()
            assert got == want and len(got) == 7 and all(len(g) == k for g in got)
        hits, counts = sh.query(qs, Metric.Cosine).per_query().take(10).collect_arrays()
        assert counts == [10] * 7 and [int(x) for x in hits["query"]] == [i for i in range(7) for _ in range(10)]
    finally:
        dist.destroy_process_group()
