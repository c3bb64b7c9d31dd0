"""CPU, world_size 2, gloo: the multi-GPU path's sharding, candidate exchange and merge.
Each rank scores its contiguous chunk-range shard (here with the ORACLE standing in for the
GPU scorer, since this box has no GPU), all-gathers fixed-size sentinel-padded candidate
blocks exactly as ShardedVecStore does, merges, and the result must equal the oracle on the
whole corpus."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, dim, cs, k, take, metric, q_out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import oracle as O
    from otters_amd._native import HIT_DTYPE
    from otters_amd.dist import gather_candidates, merge_candidates_host, pack_candidates, shard_ranges
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base, cnt = shard_ranges(n, cs, world)[rank]
    rows = O.rand_rows(base, cnt, dim, 77)           # this rank's shard of the global corpus
    queries = np.random.default_rng(5).uniform(-1, 1, (3, dim)).astype(np.float32)
    local = O.vec_query(rows, queries, metric, take, k, ties=O.TIES_CANONICAL)
    local["index"] += base                            # global row = shard base + local (src/meta_compute.rs:185)
    block = pack_candidates(local, k)
    gathered = gather_candidates(dist, torch.from_numpy(block.view(np.uint8).copy()))
    lists = gathered.numpy().view(HIT_DTYPE).reshape(world, k)
    merged = merge_candidates_host(lists, take, k)
    # PER_QUERY mode: one block of [nq, k] slots per rank, merged per query
    from otters_amd.dist import merge_candidates_host_grouped, pack_candidates_grouped
    groups = []
    for qi in range(queries.shape[0]):
        g = O.vec_query(rows, queries[qi], metric, take, k, ties=O.TIES_CANONICAL)
        g["index"] += base
        g["query"] = qi
        groups.append(g)
    gblock = pack_candidates_grouped(groups, k)
    ggath = gather_candidates(dist, torch.from_numpy(gblock.view(np.uint8).reshape(-1).copy()))
    glists = ggath.numpy().view(HIT_DTYPE).reshape(world, queries.shape[0], k)
    gmerged = merge_candidates_host_grouped(glists, take, k)
    if rank == 0:
        q_out.put((merged.tobytes(), [g.tobytes() for g in gmerged]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric,take,world", [(0, 1, 2), (1, 0, 2), (2, 1, 2), (0, 1, 4)])
def test_sharded_topk_equals_global(oracle, metric, take, world):
    import torch.multiprocessing as mp
    from otters_amd._native import HIT_DTYPE
    n, dim, cs, k = 5000, 24, 300, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, dim, cs, k, take, metric, q)) for r in range(world)]
    for p in procs:
        p.start()
    got_bytes, grouped_bytes = q.get(timeout=120)
    got = np.frombuffer(got_bytes, dtype=HIT_DTYPE)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = oracle.rand_rows(0, n, dim, 77)
    queries = np.random.default_rng(5).uniform(-1, 1, (3, dim)).astype(np.float32)
    ref = oracle.vec_query(rows, queries, metric, take, k, ties=oracle.TIES_CANONICAL)
    assert np.array_equal(got["index"], ref["index"])
    assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
    assert np.array_equal(got["query"], ref["query"])
    for qi, gb in enumerate(grouped_bytes):
        g = np.frombuffer(gb, dtype=HIT_DTYPE)
        r = oracle.vec_query(rows, queries[qi], metric, take, k, ties=oracle.TIES_CANONICAL)
        assert np.array_equal(g["index"], r["index"])
        assert np.array_equal(g["score"].view(np.uint32), r["score"].view(np.uint32))
        assert np.all(g["query"] == qi)


def test_shard_ranges_cover_corpus():
    from otters_amd.dist import shard_ranges
    for n, cs, w in ((40_000_000, 4096, 8), (5000, 300, 2), (10, 3, 4), (7, 100, 3)):
        r = shard_ranges(n, cs, w)
        assert r[0][0] == 0 and sum(c for _, c in r) == n
        for (b0, c0), (b1, _) in zip(r, r[1:]):
            assert b0 + c0 == b1
        assert all(b % cs == 0 for b, c in r if c)


def _comm_worker(rank, world, port, q_out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from otters_amd.dist import Comm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = Comm.from_torch(dist, transport="host")  # ott_comm with the HOST transport: the C ABI calls back into gloo
    assert comm.transport == "host" and comm.rank == rank and comm.world == world
    a = comm.all_gather_host(np.arange(6, dtype=np.int64).reshape(2, 3) + 100 * rank)
    b = comm.all_gather_bytes(b"x" * (3 + 5 * rank))       # variable sizes: sizes first, then padded payloads
    c = comm.all_gather_bytes(b"")                          # nothing from anyone
    comm.barrier()
    if rank == 0:
        q_out.put((a.tolist(), b, c))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_host_transport_comm_two_ranks_gloo():
    """The C ABI's ott_comm with the host-callback transport, world size 2 over gloo on CPU: control-data all-gathers
    (fixed and variable size) come back in rank order on every rank — the plumbing ott_query_sharded uses for k > 512
    and ShardedMetaStore uses for the materialised cells."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    a, b, c = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert a == [[[0, 1, 2], [3, 4, 5]], [[100, 101, 102], [103, 104, 105]]]
    assert b == [b"xxx", b"x" * 8] and c == [b"", b""]
