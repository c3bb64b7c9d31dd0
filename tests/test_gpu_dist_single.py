"""GPU: the sharded path behind the C ABI (ott_query_sharded: score -> all-gather -> merge on one stream).
 * one rank, RCCL transport (ncclCommInitRank / ncclAllGather through dlopen'ed librccl): no torch.distributed anywhere;
 * two ranks sharing the box's one GPU, HOST transport (RCCL refuses duplicate devices of one host; gloo carries the blocks
   through the ott_comm callback): real shards, kernels and merges at world size 2, including k > 512 and the default take;
 * the same two-rank runs (and world 4, and bench.py under the driver's launch line) over RCCL itself: with one NCCL_HOSTID
   per rank the ranks look like one-GPU nodes, ncclCommInitRank(world) and ncclAllGather run for real (socket transport).
Checked against the plain single-store query and the oracle.  (CPU-side N > 1 logic: tests/test_dist_cpu.py.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_store_single_rank_rccl(oracle):
    from otters_amd import Cmp, Metric, VecStore
    from otters_amd.dist import Comm, ShardedVecStore
    comm = Comm.rccl(Comm.unique_id(), 0, 1, 0)  # a real RCCL communicator of one rank; torch.distributed is not involved
    try:
        assert comm.transport == "rccl"
        assert comm.all_gather_host(np.arange(3, dtype=np.int32)).tolist() == [[0, 1, 2]]
        n, dim, base = 30000, 64, 5_000_000
        store = VecStore(dim)
        store.set_base_offset(base)
        store.append_random(n, seed=3)
        rows = oracle.rand_rows(base, n, dim, 3)
        sh = ShardedVecStore(store, comm)
        assert sh.len() == n
        q = np.random.default_rng(2).uniform(-1, 1, (2, dim)).astype(np.float32)
        for metric, k in ((Metric.Cosine, 10), (Metric.Euclidean, 100), (Metric.DotProduct, 130), (Metric.Cosine, 300),
                          (Metric.Cosine, 700)):  # 300: 8 list entries per lane; 700: beyond the device lists (host merge)
            got = sh.query(q, metric).take(k).collect()
            want = store.query(q, metric).take(k).collect()
            assert got == want and len(got) == k
            assert store.last_stats["vectors_compared"] == 2 * n
            ref = oracle.vec_query(rows, q, int(metric), 0 if metric == Metric.Euclidean else 1, k, ties=oracle.TIES_CANONICAL)
            assert [r.index - base for r in got] == [int(i) for i in ref["index"]]
            assert np.array_equal(np.array([r.score for r in got], np.float32).view(np.uint32), ref["score"].view(np.uint32))
        got = sh.query(q[0], Metric.Cosine).filter(0.9, Cmp.Gt).take(5).collect()
        assert got == []
        # the reference's default take (no .take(): every row, src/vec.rs:213) through the sharded path
        small = VecStore(dim)
        small.append_random(900, seed=5)
        shs = ShardedVecStore(small, comm)
        got = shs.query(q[0], Metric.DotProduct).collect()
        assert got == small.query(q[0], Metric.DotProduct).collect() and len(got) == 900
        # PER_QUERY through the same exchange: [nq, k] blocks, grouped device merge
        qs = np.random.default_rng(4).uniform(-1, 1, (7, dim)).astype(np.float32)
        for k in (10, 100, 600):
            got = sh.query(qs, Metric.Cosine).per_query().take(k).collect()
            want = store.query(qs, Metric.Cosine).per_query().take(k).collect()
            assert got == want and len(got) == 7 and all(len(g) == k for g in got)
        hits, counts = sh.query(qs, Metric.Cosine).per_query().take(10).collect_arrays()
        assert counts == [10] * 7 and [int(x) for x in hits["query"]] == [i for i in range(7) for _ in range(10)]
        # batches take the matrix-core cascade on the shard, then the same exchange
        qb = np.random.default_rng(9).uniform(-1, 1, (40, dim)).astype(np.float32)
        from otters_amd import Path
        got = sh.query(qb, Metric.Cosine).take(50).with_path(Path.Mfma).collect()
        assert store.last_stats["path_used"] == 2
        assert got == store.query(qb, Metric.Cosine).take(50).with_path(Path.Exact).collect()
    finally:
        comm.close()


def test_sharded_query_from_two_threads_of_one_process(oracle):
    """The SPMD entry point from THREADS of one process: two stores (two shards of one corpus, both on GPU 0), two host-transport
    comms whose all-gather is a rendezvous between the threads, ott_query_sharded called from both at once — thread-local error
    state, per-call contexts, one collective at a time per comm.  (A single-process host would use ott_store_create_multi
    instead; this is the other way its threads could drive several GPUs.)"""
    import threading
    from otters_amd import Metric, VecStore
    from otters_amd.dist import Comm, ShardedVecStore, shard_ranges
    n, dim, world = 30_000, 40, 2
    rows = oracle.rand_rows(0, n, dim, 19)
    qs = np.random.default_rng(3).uniform(-1, 1, (4, dim)).astype(np.float32)
    barrier = threading.Barrier(world)
    slots = [None] * world
    lock = threading.Lock()

    def make_allgather(rank):
        def allgather(b: bytes) -> bytes:
            with lock:
                slots[rank] = b
            barrier.wait(timeout=60)
            out = b"".join(slots)
            barrier.wait(timeout=60)  # nobody overwrites a slot before everyone has read
            return out
        return allgather
    results, errs = [None] * world, []

    def worker(rank):
        try:
            base, cnt = shard_ranges(n, 8, world)[rank]
            store = VecStore(dim)
            store.set_base_offset(base)
            store.append_random(cnt, 19)
            sh = ShardedVecStore(store, Comm.host(rank, world, make_allgather(rank)))
            out = []
            for metric, k in ((Metric.Cosine, 10), (Metric.Euclidean, 100), (Metric.DotProduct, 600)):
                hits, _ = sh.query(qs, metric).take(k).collect_arrays()
                out.append(hits.copy())
                hits, counts = sh.query(qs, metric).per_query().take(7).collect_arrays()
                out.append(hits.copy())
            results[rank] = out
            sh.comm.close()
            store.close()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, e))
            barrier.abort()
    ths = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join(timeout=120) for t in ths]
    assert not errs, errs
    i = 0
    for metric, k in ((0, 10), (1, 100), (2, 600)):
        take = 0 if metric == 1 else 1
        ref = oracle.vec_query(rows, qs, metric, take, k, ties=oracle.TIES_CANONICAL)
        for r in range(world):
            got = results[r][i]
            assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
        for r in range(world):
            got = results[r][i + 1]
            for qi in range(4):
                rq = oracle.vec_query(rows, qs[qi], metric, take, 7, ties=oracle.TIES_CANONICAL)
                g = got[qi * 7:(qi + 1) * 7]
                assert np.array_equal(g["index"], rq["index"]) and np.array_equal(g["score"].view(np.uint32), rq["score"].view(np.uint32))
        i += 2


def test_rows_appended_on_one_rank_only_between_two_sharded_queries(oracle):
    """Round 5 (advisor, high): whether a sharded query issues the small layout gather used to follow RANK-LOCAL state — a rank
    whose own row count had changed re-gathered, its peers went straight to the candidate exchange: mismatched collectives.  Now
    the layout words ride in the header of every exchange and all ranks update their table together.  Two ranks (threads, host
    transport, both shards on GPU 0); between queries ONLY the last rank appends rows; every query — k on both sides of 512,
    canonical and reference tie order (whose protocol reads the table: total rows, first row) — must return the oracle's hits
    over the corpus as it then is, on both ranks, with no collective left hanging (the transport's rendezvous would time out)."""
    import threading
    from otters_amd import Metric, VecStore
    from otters_amd.dist import Comm, ShardedVecStore, shard_ranges
    n, extra, dim, world = 20_000, 4_000, 24, 2
    rows = oracle.rand_rows(0, n + 2 * extra, dim, 23)
    qs = np.random.default_rng(8).uniform(-1, 1, (3, dim)).astype(np.float32)
    barrier = threading.Barrier(world)
    slots = [None] * world
    lock = threading.Lock()

    def make_allgather(rank):
        def allgather(b: bytes) -> bytes:
            with lock:
                slots[rank] = b
            barrier.wait(timeout=60)
            assert len({len(x) for x in slots}) == 1, "mismatched collectives: the ranks contribute blocks of different sizes"
            out = b"".join(slots)
            barrier.wait(timeout=60)
            return out
        return allgather
    results, errs = [[] for _ in range(world)], []
    (b0, c0), (b1, c1) = shard_ranges(n, 8, world)

    def worker(rank, tie):
        try:
            base, cnt = (b0, c0) if rank == 0 else (b1, c1)
            store = VecStore(dim)
            store.set_tie_order(tie)
            store.set_base_offset(base)
            store.append_random(cnt, 23)
            comm = Comm.host(rank, world, make_allgather(rank))
            for step in range(3):  # 0: as loaded; 1, 2: the LAST rank alone has appended `extra` rows more
                if step and rank == world - 1:
                    store.append_random(extra, 23)
                sh = ShardedVecStore(store, comm, global_rows=n + step * extra)
                for metric, k in ((Metric.Cosine, 10), (Metric.DotProduct, 600)):
                    hits, _ = sh.query(qs, metric).take(k).collect_arrays()
                    results[rank].append((step, int(metric), k, hits.copy()))
            comm.close()
            store.close()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            barrier.abort()
    for tie, oties in (("canonical", oracle.TIES_CANONICAL), ("reference", oracle.TIES_LITERAL)):
        for r in range(world):
            results[r].clear()
        barrier.reset()
        ths = [threading.Thread(target=worker, args=(r, tie)) for r in range(world)]
        [t.start() for t in ths]
        [t.join(timeout=180) for t in ths]
        assert not errs, errs
        assert len(results[0]) == len(results[1]) == 6
        for r in range(world):
            for step, metric, k, got in results[r]:
                ref = oracle.vec_query(rows[: n + step * extra], qs, metric, 1, k, ties=oties)
                assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32)), (tie, r, step, metric, k)
                if tie == "canonical":
                    assert np.array_equal(got["index"], ref["index"]), (tie, r, step, metric, k)
                else:
                    assert sorted(zip(got["index"].tolist(), got["query"].tolist())) == sorted(zip(ref["index"].tolist(), ref["query"].tolist())), (tie, r, step, metric, k)


def test_tie_order_changed_on_one_rank_only_between_two_sharded_queries(oracle):
    """Round 6 (advisor): which protocol a sharded query runs — the canonical exchange or the reference tie orders' (other k, other
    block sizes) — used to follow the RANK-LOCAL tie_order option: a rank that changed it between two queries issued an all-gather
    its peers did not.  Now the branch follows the table every rank holds; the changed option travels in the exchange header.  One
    rank alone switches to the reference order: EVERY rank must fail that call with the same "different tie_order options" error
    (no hang, no garbage); when the other rank follows, the next call returns the reference's outcome; switching back together
    returns the canonical one.  Two ranks as threads over the host transport, both shards on GPU 0."""
    import threading
    from otters_amd import Metric, OttersError, VecStore
    from otters_amd.dist import Comm, ShardedVecStore, shard_ranges
    n, dim, world = 16_000, 16, 2
    rows = np.round(oracle.rand_rows(0, n, dim, 5) * 4) / 4  # quantised: exact score ties, so the two orders differ
    qs = np.round(np.random.default_rng(3).uniform(-1, 1, (2, dim)) * 4).astype(np.float32) / 4
    barrier = threading.Barrier(world)
    slots = [None] * world
    lock = threading.Lock()

    def make_allgather(rank):
        def allgather(b: bytes) -> bytes:
            with lock:
                slots[rank] = b
            barrier.wait(timeout=60)
            assert len({len(x) for x in slots}) == 1, "mismatched collectives: the ranks contribute blocks of different sizes"
            out = b"".join(slots)
            barrier.wait(timeout=60)
            return out
        return allgather
    results, errs = [dict() for _ in range(world)], []
    ranges = shard_ranges(n, 8, world)

    def worker(rank):
        try:
            base, cnt = ranges[rank]
            store = VecStore(dim)
            store.set_base_offset(base)
            store.add_vectors(rows[base:base + cnt])
            comm = Comm.host(rank, world, make_allgather(rank))
            sh = ShardedVecStore(store, comm, global_rows=n)
            run = lambda: sh.query(qs, Metric.DotProduct).take(40).collect_arrays()[0].copy()
            results[rank]["canonical"] = run()
            if rank == 1:
                store.set_tie_order("reference")        # ONE rank alone
            try:
                run()
                results[rank]["lonely"] = "no error"
            except OttersError as e:
                results[rank]["lonely"] = str(e)
            if rank == 0:
                store.set_tie_order("reference")        # the other follows
            results[rank]["reference"] = run()
            store.set_tie_order("canonical")            # together
            results[rank]["canonical_again"] = run()
            comm.close()
            store.close()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            barrier.abort()
    ths = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join(timeout=180) for t in ths]
    assert not errs, errs
    ref_c = oracle.vec_query(rows, qs, oracle.METRIC_DOT, 1, 40, ties=oracle.TIES_CANONICAL)
    ref_l = oracle.vec_query(rows, qs, oracle.METRIC_DOT, 1, 40, ties=oracle.TIES_LITERAL)
    assert not np.array_equal(ref_c["index"], ref_l["index"])  # the corpus does tie at the cut
    for r in range(world):
        assert "different tie_order options" in results[r]["lonely"], (r, results[r]["lonely"])
        for key in ("canonical", "canonical_again"):
            assert np.array_equal(results[r][key]["index"], ref_c["index"]) and np.array_equal(results[r][key]["query"], ref_c["query"]), (r, key)
        got = results[r]["reference"]
        assert np.array_equal(got["score"].view(np.uint32), ref_l["score"].view(np.uint32)), r
        assert sorted(zip(got["index"].tolist(), got["query"].tolist())) == sorted(zip(ref_l["index"].tolist(), ref_l["query"].tolist())), r


def _rank_is_its_own_host(rank):
    """RCCL refuses two ranks on one device of one HOST; the host is a hash NCCL_HOSTID overrides.  One id per rank: the ranks
    look like one-GPU nodes and RCCL connects them through its socket transport (set before RCCL is first touched)."""
    import os
    os.environ.update(NCCL_HOSTID=f"ott-test-rank-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1")


def _two_rank_worker(rank, world, port, n, dim, cs, q_out, transport="host"):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if transport == "rccl":
        _rank_is_its_own_host(rank)
    import numpy as np
    import torch
    import torch.distributed as dist
    from otters_amd import Metric, VecStore
    from otters_amd.dist import Comm, ShardedVecStore, shard_ranges
    torch.cuda.set_device(0)  # both ranks share the box's one GPU: gloo carries the candidate blocks, or RCCL's socket transport
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base, cnt = shard_ranges(n, cs, world)[rank]
    store = VecStore(dim)
    store.set_base_offset(base)
    store.append_random(cnt, seed=11)  # counter-based generator keyed by GLOBAL row: the shards tile one corpus
    # gloo group -> ott_comm with the HOST transport, or an RCCL communicator whose id travelled over the group
    sh = ShardedVecStore(store, dist if transport == "host" else Comm.from_torch(dist, 0, transport="rccl"))  # the shard sizes are exchanged once
    assert sh.comm.transport == transport and sh.len() == n
    qs = np.random.default_rng(6).uniform(-1, 1, (5, dim)).astype(np.float32)
    out = {}
    for metric in (Metric.Cosine, Metric.Euclidean, Metric.DotProduct):
        hits, _ = sh.query(qs, metric).take(40).collect_arrays()
        out[("merged", int(metric))] = hits.tobytes()
        hits, counts = sh.query(qs, metric).per_query().take(12).collect_arrays()
        out[("perq", int(metric))] = (hits.tobytes(), counts)
    hits, _ = sh.query(qs[:2], Metric.Cosine).take(1500).collect_arrays()     # k > 512: whole lists exchanged, host merge
    out["large_k"] = hits.tobytes()
    hits, _ = sh.query(qs[0], Metric.DotProduct).collect_arrays()              # default take = every row of the CORPUS
    out["default_take"] = hits.tobytes()
    # config 4's shape in small: 1024 queries, cosine top-100 per query — each shard runs the matrix-core cascade (four
    # 256-query blocks per row tile), the exchange carries [1024][128] slots per rank, one grouped device merge
    from otters_amd import Path
    big = np.random.default_rng(12).uniform(-1, 1, (1024, dim)).astype(np.float32)
    hits, counts = sh.query(big, Metric.Cosine).per_query().take(100).with_path(Path.Mfma).collect_arrays()
    assert store.last_stats["path_used"] == 2 and counts == [100] * 1024
    out["c4_shape"] = hits.tobytes()
    # the reference's outcome at exact score ties across shards (tie_order = 1: ONE collector over the whole corpus): quantised
    # rows, so nearly every cut runs through a group of equal scores that spans both shards
    rng = np.random.default_rng(404)
    tn, tdim = 4000, 6
    trows = rng.integers(-2, 3, (tn, tdim)).astype(np.float32)
    tq = rng.integers(-2, 3, (3, tdim)).astype(np.float32)
    tq[np.all(tq == 0, axis=1)] = 1.0
    tbase, tcnt = shard_ranges(tn, 8, world)[rank]
    tstore = VecStore(tdim)
    tstore.set_tie_order("reference")
    tstore.set_base_offset(tbase)
    tstore.add_vectors(trows[tbase:tbase + tcnt])
    tsh = ShardedVecStore(tstore, sh.comm)
    ties = {}
    for k in (1, 7, 20, 64, 150, 700):
        hits, _ = tsh.query(tq, Metric.DotProduct).take(k).collect_arrays()
        ties[("merged", k)] = hits.tobytes()
        hits, counts = tsh.query(tq, Metric.Euclidean).per_query().take(k).collect_arrays()
        ties[("perq", k)] = (hits.tobytes(), counts)
    out["ties"] = ties
    # MetaStore semantics across ranks (tie_order = 2: one collector per chunk, concat-sort-truncate, src/meta.rs:678-709): the
    # chunks that hold candidates of an ambiguous cut are re-queried by the ranks that own them
    tcs = 40
    mbase, mcnt = shard_ranges(tn, tcs, world)[rank]
    mstore = VecStore(tdim)
    mstore.set_chunk_size(tcs)
    mstore.set_tie_order("reference_chunked")
    mstore.set_base_offset(mbase)
    mstore.add_vectors(trows[mbase:mbase + mcnt])
    msh = ShardedVecStore(mstore, sh.comm)
    mties = {}
    for k in (1, 4, 10, 30, 100):
        hits, _ = msh.query(tq, Metric.DotProduct).take(k).collect_arrays()
        mties[("dot", k)] = hits.tobytes()
        hits, _ = msh.query(tq[0], Metric.Euclidean).take(k).collect_arrays()
        mties[("l2", k)] = hits.tobytes()
    out["meta_ties"] = mties
    # a layout the reference's tie order cannot be reproduced on — a shard that does not start a multiple of 8 rows after the
    # first — is refused on EVERY rank, by the same check, before any candidate exchange (no rank-local protocol choice)
    ubase = 0 if rank == 0 else 1003
    ucnt = 1003 if rank == 0 else tn - 1003
    ustore = VecStore(tdim)
    ustore.set_tie_order("reference")
    ustore.set_base_offset(ubase)
    ustore.add_vectors(trows[ubase:ubase + ucnt])
    ush = ShardedVecStore(ustore, sh.comm)
    try:
        ush.query(tq, Metric.DotProduct).take(7).collect_arrays()
        out["unaligned"] = "no error"
    except Exception as e:  # noqa: BLE001
        out["unaligned"] = str(e)
    ustore.set_tie_order("canonical")  # (both ranks: a collective change) the canonical order has no such requirement
    hits, _ = ush.query(tq, Metric.DotProduct).take(7).collect_arrays()
    out["unaligned_canonical"] = hits.tobytes()
    # a failure only ONE rank sees (its output buffer is too small) reaches the other with the exchange: both return an error
    # at once, nobody waits for a timeout
    import ctypes as C
    from otters_amd import _native as N
    d = N.QueryDesc()
    q1 = np.ascontiguousarray(qs[0])
    d.queries, d.nq, d.metric, d.take, d.k = q1.ctypes.data, 1, 0, 1, 20
    buf = np.zeros(32, dtype=N.HIT_DTYPE)
    n_out = C.c_uint64(0)
    import time
    t_fail = time.perf_counter()
    rc = N.lib().ott_query_sharded(store._handle(), sh.comm._h, C.byref(d), N.ptr(buf), 5 if rank == 1 else 32, C.byref(n_out), None, None)
    out["rank_failure"] = (rc, N.lib().ott_last_error().decode(), time.perf_counter() - t_fail)
    allf = sh.comm.all_gather_bytes(repr(out["rank_failure"][:2]).encode())
    out["rank_failure_all"] = [b.decode() for b in allf]
    if rank == 0:
        q_out.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["host", "rccl"])
def test_sharded_store_two_ranks_one_gpu(oracle, transport):
    """world_size 2 on the GPU: two processes, each with its own shard in HBM (device 0), real scoring / top-k / merge
    kernels, candidate blocks exchanged over gloo (host transport) or by ncclAllGather (each rank its own NCCL_HOSTID: RCCL's
    socket transport).  The merged and per-query results must equal the oracle on the whole corpus."""
    import torch.multiprocessing as mp
    from otters_amd._native import HIT_DTYPE
    n, dim, cs, world = 41_000, 48, 512, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, n, dim, cs, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows = oracle.rand_rows(0, n, dim, 11)
    qs = np.random.default_rng(6).uniform(-1, 1, (5, dim)).astype(np.float32)
    for metric in (0, 1, 2):
        take = 0 if metric == 1 else 1
        got = np.frombuffer(out[("merged", metric)], dtype=HIT_DTYPE)
        ref = oracle.vec_query(rows, qs, metric, take, 40, ties=oracle.TIES_CANONICAL)
        assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["query"], ref["query"])
        assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
        raw, counts = out[("perq", metric)]
        got = np.frombuffer(raw, dtype=HIT_DTYPE)
        assert counts == [12] * 5
        for qi in range(5):
            ref = oracle.vec_query(rows, qs[qi], metric, take, 12, ties=oracle.TIES_CANONICAL)
            g = got[qi * 12:(qi + 1) * 12]
            assert np.array_equal(g["index"], ref["index"]) and np.all(g["query"] == qi)
            assert np.array_equal(g["score"].view(np.uint32), ref["score"].view(np.uint32))
    got = np.frombuffer(out["large_k"], dtype=HIT_DTYPE)
    ref = oracle.vec_query(rows, qs[:2], 0, 1, 1500, ties=oracle.TIES_CANONICAL)
    assert got.size == 1500 and np.array_equal(got["index"], ref["index"]) and np.array_equal(got["query"], ref["query"])
    assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
    got = np.frombuffer(out["default_take"], dtype=HIT_DTYPE)
    ref = oracle.vec_query(rows, qs[0], 2, 1, n, ties=oracle.TIES_CANONICAL)
    assert got.size == n and np.array_equal(got["index"], ref["index"])
    assert np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
    # reference tie order across two shards == the literal collector over the whole corpus (sets and score sequences)
    trng = np.random.default_rng(404)
    trows = trng.integers(-2, 3, (4000, 6)).astype(np.float32)
    tq = trng.integers(-2, 3, (3, 6)).astype(np.float32)
    tq[np.all(tq == 0, axis=1)] = 1.0
    for k in (1, 7, 20, 64, 150, 700):
        got = np.frombuffer(out["ties"][("merged", k)], dtype=HIT_DTYPE)
        lit = oracle.vec_query(trows, tq, oracle.METRIC_DOT, oracle.TAKE_MAX, k, ties=oracle.TIES_LITERAL)
        assert np.array_equal(got["score"].view(np.uint32), lit["score"].view(np.uint32)), ("ties merged", k)
        assert sorted(zip(got["index"].tolist(), got["query"].tolist())) == sorted(zip(lit["index"].tolist(), lit["query"].tolist())), ("ties merged", k)
        raw, counts = out["ties"][("perq", k)]
        got = np.frombuffer(raw, dtype=HIT_DTYPE)
        o = 0
        for qi in range(3):
            lit = oracle.vec_query(trows, tq[qi], oracle.METRIC_EUCLIDEAN, oracle.TAKE_MIN, k, ties=oracle.TIES_LITERAL)
            g = got[o:o + counts[qi]]
            assert np.array_equal(g["score"].view(np.uint32), lit["score"].view(np.uint32)) and sorted(g["index"].tolist()) == sorted(lit["index"].tolist()), ("ties perq", k, qi)
            o += counts[qi]
    for k in (1, 4, 10, 30, 100):
        for name, metric, take, qq in (("dot", oracle.METRIC_DOT, oracle.TAKE_MAX, tq), ("l2", oracle.METRIC_EUCLIDEAN, oracle.TAKE_MIN, tq[0])):
            got = np.frombuffer(out["meta_ties"][(name, k)], dtype=HIT_DTYPE)
            lit, _ = oracle.meta_query(trows, 40, qq, metric, take, k, ties=oracle.TIES_LITERAL)
            assert np.array_equal(got["score"].view(np.uint32), lit["score"].view(np.uint32)), ("meta ties", name, k)
            assert sorted(got["index"].tolist()) == sorted(lit["index"].tolist()), ("meta ties", name, k)
    assert "multiple of 8 rows" in out["unaligned"], out["unaligned"]
    got = np.frombuffer(out["unaligned_canonical"], dtype=HIT_DTYPE)
    ref = oracle.vec_query(trows, tq, oracle.METRIC_DOT, oracle.TAKE_MAX, 7, ties=oracle.TIES_CANONICAL)
    assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["score"].view(np.uint32), ref["score"].view(np.uint32))
    rc0, msg0, secs = out["rank_failure"]
    assert rc0 != 0 and "rank 1 failed" in msg0 and secs < 20, out["rank_failure"]  # rank 0 learns of rank 1's failure from the exchange
    assert "output capacity" in out["rank_failure_all"][1], out["rank_failure_all"]
    got = np.frombuffer(out["c4_shape"], dtype=HIT_DTYPE).reshape(1024, 100)
    big = np.random.default_rng(12).uniform(-1, 1, (1024, dim)).astype(np.float32)
    for qi in range(0, 1024, 37):  # 28 of the 1024 lists against the oracle on the whole corpus
        ref = oracle.vec_query(rows, big[qi], 0, 1, 100, ties=oracle.TIES_CANONICAL, fast=True)
        assert np.array_equal(got[qi]["index"], ref["index"]) and np.all(got[qi]["query"] == qi)
        assert np.array_equal(got[qi]["score"].view(np.uint32), ref["score"].view(np.uint32))


def _meta_corpus(n, dim, cs):
    from otters_amd import Column, DataType
    rng = np.random.default_rng(21)
    vec = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    chunk = np.arange(n) // cs
    price = (chunk % 5) * 20.0 + rng.uniform(0, 25, n)
    price_null = rng.random(n) < 0.05
    ver = ((chunk % 3) + rng.integers(0, 2, n)).astype(np.int32)
    grade = np.array(["A", "B", "C", "D"])[(chunk + rng.integers(0, 2, n)) % 4]
    grade_null = rng.random(n) < 0.03

    def cols(lo, hi):
        return [Column.from_numpy("price", DataType.Float64, price[lo:hi], price_null[lo:hi]),
                Column.from_numpy("version", DataType.Int32, ver[lo:hi]),
                Column.from_numpy("grade", DataType.String, grade[lo:hi], grade_null[lo:hi])]
    return vec, cols


def _meta_filters():
    from otters_amd import col
    return [lambda: col("price").lt(50.0) & col("version").gte(1),
            lambda: col("grade").eq("A") | col("grade").eq("C"),
            lambda: col("price").gt(70.0) & col("grade").neq("B")]


def _meta_two_rank_worker(rank, world, port, n, dim, cs, q_out, transport="host"):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if transport == "rccl":
        _rank_is_its_own_host(rank)
    import torch
    import torch.distributed as dist
    from otters_amd import Cmp, MetaStore, Metric
    from otters_amd.dist import Comm, ShardedMetaStore, shard_ranges
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vec, cols = _meta_corpus(n, dim, cs)
    base, cnt = shard_ranges(n, cs, world)[rank]
    meta = MetaStore.from_columns(cols(base, base + cnt)).with_vectors(vec[base:base + cnt]).with_chunk_size(cs).build()
    sms = ShardedMetaStore(meta, dist if transport == "host" else Comm.from_torch(dist, 0, transport="rccl"), base_row=base)
    assert sms.comm.transport == transport
    qs = np.random.default_rng(8).uniform(-1, 1, (2, dim)).astype(np.float32)
    out = []
    for fi, f in enumerate(_meta_filters()):
        plan = sms.query_batch(qs, Metric.Cosine).meta_filter(f()).vec_filter(0.0, Cmp.Gt)
        res = (plan if fi == 2 else plan.take(15)).collect()  # the last one with the reference's default take (every row)
        stt = sms.last_query_stats()
        out.append((res.indices, res.scores, {c: (res.data[c].values() if c == "grade" else res.data[c].values().tolist()) for c in res.columns},
                    {c: res.data[c].null_mask().tolist() for c in res.columns},
                    (stt.total_chunks, stt.pruned_chunks, stt.evaluated_chunks, stt.vectors_compared)))
    if rank == 0:
        q_out.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["host", "rccl"])
def test_sharded_meta_store_two_ranks_one_gpu(transport):
    """MetaStore sharded over two ranks (vectors + metadata columns per shard, local zonemap prune and GPU row masks, one
    candidate exchange, hits materialised by their owners) == the same query on one MetaStore holding everything."""
    import torch.multiprocessing as mp
    from otters_amd import Cmp, MetaStore, Metric
    n, dim, cs, world = 24_000, 32, 500, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_meta_two_rank_worker, args=(r, world, port, n, dim, cs, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    vec, cols = _meta_corpus(n, dim, cs)
    whole = MetaStore.from_columns(cols(0, n)).with_vectors(vec).with_chunk_size(cs).build()
    qs = np.random.default_rng(8).uniform(-1, 1, (2, dim)).astype(np.float32)
    for fi, (f, (idx, scores, data, nulls, stats)) in enumerate(zip(_meta_filters(), got)):
        plan = whole.query_batch(qs, Metric.Cosine).meta_filter(f()).vec_filter(0.0, Cmp.Gt)
        ref = (plan if fi == 2 else plan.take(15)).collect()
        assert idx == ref.indices and (len(idx) == 15 if fi < 2 else len(idx) > 512)
        assert np.array_equal(np.array(scores, np.float32).view(np.uint32), np.array(ref.scores, np.float32).view(np.uint32))
        for c in ref.columns:
            rn = ref.data[c].null_mask().tolist()
            assert nulls[c] == rn
            rv = ref.data[c].values() if c == "grade" else ref.data[c].values().tolist()
            assert [v for v, z in zip(data[c], rn) if not z] == [v for v, z in zip(rv, rn) if not z]
        rs = whole.last_query_stats()
        assert stats == (rs.total_chunks, rs.pruned_chunks, rs.evaluated_chunks, rs.vectors_compared)


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OTT_BENCH_CHILD")}
    env.update(kw)
    return env


def test_bench_two_ranks_through_both_launchers():
    """bench.py's N-rank launch contract on the one GPU of the test box (OTT_BENCH_SINGLE_DEVICE=1: both ranks use GPU 0 and
    the candidate blocks travel over the host transport — RCCL refuses two ranks on one device): its own launcher
    (`python bench.py --gpus 2`) and the driver's (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`)
    must each print exactly ONE JSON line, with n_gpus == 2, the parity gate passed, weak scaling, and a whole-job value that
    covers both shards."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    args = ["--gpus", "2", "--rows", "1000448", "--steps", "3", "--warmup", "1"]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launchers = {
        "self": [sys.executable, bench] + args,
        "torchrun": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                     "--master-port", str(port), bench] + args,
    }
    for name, cmd in launchers.items():
        r = subprocess.run(cmd, env=_clean_env(OTT_BENCH_SINGLE_DEVICE="1"), capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, (name, r.stdout)
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["parity_checked"] is True, (name, d)
        assert d["scaling"] == "weak" and d["config"]["rows_per_gpu"] == 1000448 and d["config"]["transport"] == "host"
        # value = bytes scanned by BOTH shards per second: twice what one rank's pass over its shard amounts to
        per_rank = 1000448 * (768 * 4 + 4) / (d["ms_per_step"] * 1e-3) / 1e9
        assert abs(d["value"] - 2 * per_rank) <= 0.02 * d["value"], (name, d["value"], per_rank)


def test_comm_create_gives_up_when_a_peer_never_arrives():
    """ott_comm_create (ncclCommInitRank) with world = 2 and nobody at rank 1: after the comm timeout the call must come back
    with an error that says so — not hang the job (a rank that died before the rendezvous used to leave its peers inside
    RCCL's bootstrap for good).  Run in a child process: the abandoned bootstrap thread dies with it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, time\n"
        f"sys.path.insert(0, {root!r})\n"
        "from otters_amd import _native as N\n"
        "from otters_amd.dist import Comm\n"
        "uid = Comm.unique_id()\n"
        "t0 = time.time()\n"
        "try:\n"
        "    Comm.rccl(uid, 0, 2, 0)\n"
        "    print('CREATED')\n"
        "except N.OttersError as e:\n"
        "    print('ERR', round(time.time() - t0, 1), str(e))\n"
        "sys.stdout.flush()\n"
        "os._exit(0)\n"
    )
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(OTT_COMM_TIMEOUT_MS="4000"), capture_output=True, text=True, timeout=180)
    out = [ln for ln in r.stdout.splitlines() if ln.startswith(("ERR", "CREATED"))]
    assert out and out[0].startswith("ERR"), (r.stdout, r.stderr[-1500:])
    assert "did not arrive" in out[0] and float(out[0].split()[1]) < 60.0, out[0]


def _check_multi_gpu_extras(d, n_gpus, rows, transport, nq=64):
    """Round 6: for N > 1 the line carries config 4's shape and the metric's corpus split N ways beside the weak-scaling headline
    (bench.py: multi_gpu_extras): fields present and consistent, parity of the batch against the exact-order path checked."""
    ex = d["extras"]
    assert "error" not in ex, ex
    c4, st = ex["config4"], ex["strong_10M"]
    rows_c4 = (min(5_000_000, rows if rows <= 5_000_000 else rows // 2) // 1024) * 1024
    assert c4["rows_per_gpu"] == rows_c4 and c4["n_gpus"] == n_gpus and c4["nq"] == nq and c4["k"] == 100, c4
    assert c4["parity_checked_queries"] == 8 and c4["transport"] == transport, c4
    for mode, hits, block in (("merged", 100, 100 * 16), ("per_query", nq * 100, nq * 100 * 16)):
        m = c4[mode]
        assert m["hits"] == hits and m["exchange_bytes_per_gpu"] == block and m["exchange_bytes_all_gpus"] == block * n_gpus, (mode, m)
        assert m["ms_per_batch"] > 0 and abs(m["queries_per_sec"] - nq / (m["ms_per_batch"] * 1e-3)) <= 0.01 * m["queries_per_sec"], (mode, m)
        assert m["merge_us"] > 0 and m["allgather_us"] >= 0, (mode, m)
    rows_s = ((rows // n_gpus) // 1024) * 1024
    assert st["scaling"] == "strong" and st["rows_per_gpu"] == rows_s and st["rows_total"] == n_gpus * rows_s and st["n_gpus"] == n_gpus, st
    assert st["ms_per_step"] > 0 and abs(st["GBs_scanned"] - st["rows_total"] * (768 * 4 + 4) / (st["ms_per_step"] * 1e-3) / 1e9) <= 0.02 * st["GBs_scanned"], st
    assert d["scaling"] == "weak"  # `value` stays the weak-scaling headline; the two readings are labelled


def test_bench_eight_ranks_one_gpu():
    """The driver's scaling run is 8 ranks; this box has one GPU.  `bench.py --gpus 8` with OTT_BENCH_SINGLE_DEVICE=1 puts all
    eight ranks on GPU 0 (host transport): eight shards with their own base offsets, eight candidate blocks gathered and
    merged per query, the launcher watching eight children — everything of the 8-rank path except RCCL's own transport.  One
    JSON line, n_gpus == 8, the parity gate (every returned score re-derived by the oracle from the regenerated GLOBAL row) passed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--rows", "200704", "--steps", "3", "--warmup", "1", "--c4-queries", "64"],
                       env=_clean_env(OTT_BENCH_SINGLE_DEVICE="1"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["parity_checked"] is True and d["scaling"] == "weak" and d["config"]["transport"] == "host", d
    assert d["parity"]["rescored_by_oracle"] == 10
    per_rank = 200704 * (768 * 4 + 4) / (d["ms_per_step"] * 1e-3) / 1e9
    assert abs(d["value"] - 8 * per_rank) <= 0.02 * d["value"]
    # the line carries its own evidence of the exchange: per-step gather / merge times on rank 0, every rank's kernel time
    ex = d["exchange"]
    assert ex["rccl"] is None and ex["allgather_us"] > 0 and ex["merge_us"] > 0, ex  # (host transport here: no RCCL communicator)
    assert 0 < ex["kernel_ms_per_rank"]["min"] <= ex["kernel_ms_per_rank"]["max"], ex
    assert d["config"]["processes"] == 8
    _check_multi_gpu_extras(d, 8, 200704, "host")


@pytest.mark.parametrize("shards", [2, 8])
def test_bench_inprocess_n_shards_one_gpu(shards):
    """`bench.py --gpus N --inprocess`: ONE process, one store over N shards (ott_store_create_multi; all on GPU 0 here), the
    same JSON line — n_gpus == N, weak scaling, parity gate passed on the global rows, the in-process sharding named."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(shards), "--inprocess", "--rows", "200704", "--steps", "3", "--warmup", "1",
                        "--c4-queries", "64"],
                       env=_clean_env(OTT_BENCH_SINGLE_DEVICE="1"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == shards and d["parity_checked"] is True and d["scaling"] == "weak", d
    assert d["config"]["processes"] == 1 and d["config"]["transport"] == "peer" and "in-process" in d["config"]["sharding"], d["config"]
    per_shard = 200704 * (768 * 4 + 4) / (d["ms_per_step"] * 1e-3) / 1e9
    assert abs(d["value"] - shards * per_shard) <= 0.02 * d["value"]
    assert d["exchange"]["merge_us"] > 0 and d["exchange"]["kernel_ms_per_rank"]["max"] > 0
    _check_multi_gpu_extras(d, shards, 200704, "peer")


@pytest.mark.parametrize("world", [2, 4])
def test_rccl_world_gt_1_on_one_gpu_through_the_socket_transport(world):
    """RCCL with world > 1, executed: each rank names itself a host of its own (NCCL_HOSTID), so RCCL's duplicate-device check
    does not apply and the ranks — all on the box's one GPU — are connected through RCCL's network transport over `lo`.
    ncclCommInitRank(world) bootstraps for real, ncclAllGather moves the candidate blocks between RCCL kernels, and every
    rank's merged result (k = 10 .. 700, merged and per query, exact ties across the shards) equals one store holding the whole
    corpus, bit for bit (benchmarks/rccl_two_ranks_one_gpu.py; everything of the N > 1 path except xGMI)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "benchmarks", "rccl_two_ranks_one_gpu.py"), str(world), "100000", "96"],
                       env=_clean_env(NCCL_DEBUG="WARN"), capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "PROBE OK" in r.stdout, (r.stdout[-3000:], r.stderr[-1000:])
    reports = [json.loads(ln.split("REPORT ", 1)[1]) for ln in r.stdout.splitlines() if "REPORT {" in ln]
    assert sorted(rep["rank"] for rep in reports) == list(range(world))
    for rep in reports:
        assert rep["transport"] == "rccl" and rep["world"] == world and rep["global_rows"] == world * 100000
        assert len(rep["cases"]) == 5 and all(c["equal"] for c in rep["cases"]), rep


def test_bench_two_ranks_one_gpu_over_rccl():
    """`bench.py --gpus 2` exactly as the driver starts it (torch.distributed.run), both ranks on GPU 0, the data plane on
    RCCL (OTT_BENCH_SINGLE_DEVICE=rccl): one JSON line, transport "rccl", parity gate passed on the global rows."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "1000448", "--steps", "3", "--warmup", "1", "--c4-queries", "64"]
    r = subprocess.run(cmd, env=_clean_env(OTT_BENCH_SINGLE_DEVICE="rccl"), capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity_checked"] is True and d["config"]["transport"] == "rccl", d
    assert "RCCL all-gather" in d["config"]["sharding"]
    # what RCCL itself reports: ncclCommCount == --gpus (bench.py exits non-zero otherwise), ncclGetVersion
    assert d["exchange"]["rccl"]["nranks"] == 2 and d["exchange"]["rccl"]["version"] >= 20000, d["exchange"]
    assert d["exchange"]["allgather_us"] > 0 and d["exchange"]["merge_us"] > 0
    _check_multi_gpu_extras(d, 2, 1000448, "rccl")
    assert d["extras"]["config4"]["rccl"]["nranks"] == 2


def test_bench_eight_ranks_one_gpu_over_rccl():
    """The 8-rank line with the data plane on RCCL (OTT_BENCH_SINGLE_DEVICE=rccl: eight ranks on GPU 0, RCCL's socket transport between
    them): ncclCommCount == 8 in the headline's exchange AND in the config-4 extras, whose per-query block is what config 4 all-gathers
    (src/meta.rs:678-709 is what that exchange stands for)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--rows", "200704", "--steps", "3", "--warmup", "1", "--c4-queries", "64"],
                       env=_clean_env(OTT_BENCH_SINGLE_DEVICE="rccl"), capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["parity_checked"] is True and d["config"]["transport"] == "rccl", d
    assert d["exchange"]["rccl"]["nranks"] == 8
    _check_multi_gpu_extras(d, 8, 200704, "rccl")
    assert d["extras"]["config4"]["rccl"]["nranks"] == 8


@pytest.mark.parametrize("walk", [0, 1], ids=["rank", "walk"])
def test_merge_hits_device_against_a_host_sort(walk):
    """ott_merge_hits_device / _grouped on their own (what follows the all-gather of a sharded query): lists of sorted hits with
    sentinels behind them, 1 .. 64 lists x 1 .. 5 groups x list lengths 64 .. 512, quantised scores (ties across lists), NaN
    scores, empty lists, and a plateau wide enough to overflow the rank kernel's buffer (it then falls back to the insertion
    merge inside the same launch).  Expected: better score first, then lower list, then lower position."""
    import ctypes as C
    import torch
    from otters_amd import VecStore
    from otters_amd import _native as N
    store = VecStore(8)
    store.add_vectors(np.ones((4, 8), np.float32))
    store.set_option("force_fallback", 1 if walk else 0)  # bit 1: the insertion merge
    L = N.lib()
    rng = np.random.default_rng(99)
    cases = [(nl, ng, ll, k) for nl in (1, 2, 3, 8, 17, 64) for ng in (1, 5) for ll, k in ((64, 10), (128, 100), (256, 200), (512, 512), (128, 1000))]
    cases.append((64, 1, 512, 300))  # plateau
    for ci, (n_lists, n_groups, list_len, k) in enumerate(cases):
        plateau = ci == len(cases) - 1
        for take in (0, 1):
            lists = np.zeros((n_lists, n_groups, list_len), dtype=N.HIT_DTYPE)
            lists["index"] = np.uint64(0xFFFFFFFFFFFFFFFF)
            lists["score"] = np.float32(np.nan)
            lists["query"] = 0xFFFFFFFF
            for li in range(n_lists):
                for g in range(n_groups):
                    cnt = list_len if plateau else int(rng.choice([0, 1, list_len // 3, list_len]))
                    sc = np.full(cnt, 0.5, np.float32) if plateau else (rng.integers(-6, 7, cnt) / 4).astype(np.float32)
                    sc = np.sort(sc)[::-1] if take == 1 else np.sort(sc)
                    if cnt > 3 and not plateau:
                        sc[1] = np.nan  # a NaN inside a list is skipped, not a terminator
                    lists["score"][li, g, :cnt] = sc
                    lists["index"][li, g, :cnt] = (li * 1_000_000 + g * 10_000 + np.arange(cnt)).astype(np.uint64)
                    lists["query"][li, g, :cnt] = g
            dev = torch.from_numpy(lists.view(np.uint8).reshape(-1).copy()).cuda()
            pool = n_lists * list_len
            out = np.zeros(n_groups * min(k, pool), dtype=N.HIT_DTYPE)
            n_out = C.c_uint64(0)
            per = (C.c_uint64 * n_groups)()
            N.check(L.ott_merge_hits_device_grouped(store._handle(), C.c_void_p(dev.data_ptr()), n_lists, n_groups, list_len, take, k,
                                                    N.ptr(out), C.byref(n_out), per))
            o = 0
            for g in range(n_groups):
                cand = []
                for li in range(n_lists):
                    for pos in range(list_len):
                        h = lists[li, g, pos]
                        if h["index"] != np.uint64(0xFFFFFFFFFFFFFFFF) and not np.isnan(h["score"]):
                            cand.append((float(h["score"]), li * list_len + pos, int(h["index"])))
                cand.sort(key=lambda t: (-t[0] if take == 1 else t[0], t[1]))
                want = cand[:k]
                got = out[o:o + per[g]]
                ctx = (walk, n_lists, n_groups, list_len, k, take, g)
                assert per[g] == len(want), ctx
                assert got["index"].tolist() == [w[2] for w in want], ctx
                assert np.array_equal(got["score"], np.array([w[0] for w in want], np.float32)), ctx
                o += per[g]
            assert n_out.value == o
            if n_groups == 1:  # the ungrouped entry point
                out1 = np.zeros(min(k, pool), dtype=N.HIT_DTYPE)
                n1 = C.c_uint64(0)
                N.check(L.ott_merge_hits_device(store._handle(), C.c_void_p(dev.data_ptr()), n_lists, list_len, take, k, N.ptr(out1), C.byref(n1)))
                assert n1.value == o and np.array_equal(out1[:o]["index"], out[:o]["index"])
    store.close()
