"""Multi-GPU: the corpus sharded by contiguous chunk ranges, one process per GPU.

The reference fans chunks out over a rayon pool and then concat-sort-truncates the per-chunk
top-k lists (src/meta.rs:678-709).  Here rank g owns rows [base_g, base_g + n_g) of the global
corpus in its own HBM; a query is scored on every shard independently (no data-path
collective), then ONE exchange — an all-gather of the fixed-size per-GPU top-k candidate
lists over RCCL/xGMI (k x 16 B per GPU: latency-bound, far below link bandwidth) — and the
same final merge kernel on every rank.

All of that lives behind the C ABI (`ott_query_sharded`, include/otters_hip.h): score -> ncclAllGather
-> merge are queued on one HIP stream with no host synchronisation in between.  This module is a thin
caller: `Comm` wraps an `ott_comm` (RCCL, or a host-callback transport for two test ranks on one GPU and
for CPU tests), `ShardedVecStore` / `ShardedMetaStore` mirror the single-GPU classes.  `torch.distributed`,
when used at all, only bootstraps (it carries the 128-byte RCCL id) or backs the host-callback transport.

The host-side reference merge (`pack_candidates`, `gather_candidates`, `merge_candidates_host`) stays as
plain functions: the CPU tests use them as the checker of the device exchange.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Callable, Optional

import numpy as np

from . import _native as N
from .meta import MetaQueryPlan, MetaQueryResults, MetaQueryStats
from .vec import Metric, Mode, ResolvedQuery, SearchResult, VecQueryPlan, VecStore

SENTINEL_INDEX = np.uint64(0xFFFFFFFFFFFFFFFF)


def shard_ranges(n_rows: int, chunk_size: int, world: int):
    """Contiguous chunk ranges: rank g owns chunks [g*C/G, (g+1)*C/G) -> (base_row, n_rows) per rank."""
    n_chunks = (n_rows + chunk_size - 1) // chunk_size
    out = []
    for g in range(world):
        c0, c1 = g * n_chunks // world, (g + 1) * n_chunks // world
        r0, r1 = min(c0 * chunk_size, n_rows), min(c1 * chunk_size, n_rows)
        out.append((r0, r1 - r0))
    return out


def pack_candidates(hits: np.ndarray, cap: int) -> np.ndarray:
    """Fixed-size candidate block for the all-gather: `cap` ott_hit slots, sentinel padded."""
    buf = np.zeros(cap, dtype=N.HIT_DTYPE)
    buf["index"] = SENTINEL_INDEX
    buf["score"] = np.float32(np.nan)
    buf["query"] = 0xFFFFFFFF
    buf[: hits.size] = hits[:cap]
    return buf


def gather_candidates(dist, local_bytes):
    """all_gather of equal-size candidate blocks (uint8 tensors; RCCL on GPUs, gloo on CPU)."""
    import torch
    out = torch.empty(dist.get_world_size() * local_bytes.numel(), dtype=torch.uint8, device=local_bytes.device)
    dist.all_gather_into_tensor(out, local_bytes)
    return out


def merge_candidates_host(lists: np.ndarray, take: int, k: int) -> np.ndarray:
    """Reference-order merge of gathered candidate lists on the host (used by the gloo/CPU
    tests as the checker of the device merge; src/meta.rs:699-709).  lists: [world, cap] HIT_DTYPE."""
    flat = lists.reshape(-1)
    real = flat[flat["index"] != SENTINEL_INDEX]
    bits = real["score"].view(np.uint32).astype(np.uint64)
    key = np.where(bits & 0x80000000, ~bits & 0xFFFFFFFF, bits | 0x80000000)
    if take == 1:
        key = 0xFFFFFFFF - key
    order = np.lexsort((real["query"], real["index"], key))
    return real[order][:k]


def pack_candidates_grouped(groups, cap: int) -> np.ndarray:
    """PER_QUERY block for the all-gather: [n_groups, cap] slots, each group sentinel padded."""
    return np.stack([pack_candidates(g, cap) for g in groups]) if groups else np.zeros((0, cap), dtype=N.HIT_DTYPE)


def merge_candidates_host_grouped(lists: np.ndarray, take: int, k: int):
    """Per-query merge of gathered PER_QUERY blocks.  lists: [world, n_groups, cap] HIT_DTYPE -> one array per group."""
    return [merge_candidates_host(lists[:, g, :], take, k) for g in range(lists.shape[1])]


import weakref  # noqa: E402


def _destroy_comm(handle) -> None:
    """ott_comm_destroy of a raw handle (the finaliser of a Comm: runs when the object is collected, and at interpreter exit
    BEFORE module teardown — weakref.finalize objects are called by atexit while the HIP runtime is still up)."""
    try:
        N.lib().ott_comm_destroy(handle)
    except Exception:  # noqa: BLE001 -- at exit: nothing sensible to do with it
        pass


class Comm:
    """An `ott_comm` (include/otters_hip.h): the candidate exchange of sharded queries.

    Comm.rccl(uid, rank, world, device)   ncclCommInitRank (uid = Comm.unique_id() of rank 0, handed over out of band)
    Comm.host(rank, world, allgather)     host transport: `allgather(bytes) -> bytes` (rank-order concatenation)
    Comm.from_torch(dist, device, ...)    bootstrap from an initialised torch.distributed group
    """

    def __init__(self, handle, rank: int, world: int, keep=None):
        self._h = handle
        self.rank, self.world = int(rank), int(world)
        self._keep = keep  # the ctypes callback object must outlive the comm
        # a comm that is dropped without close() is destroyed when it is collected (RCCL communicator, stream, device and pinned
        # buffers: a host that makes one comm per job must not leak one per job), and at exit at the latest
        self._fin = weakref.finalize(self, _destroy_comm, handle)

    def set_timeout_ms(self, ms: int) -> None:
        """How long a collective (or the RCCL rendezvous) may wait for a missing peer before it fails with an error
        (ott_comm_set_timeout_ms; 0 = for ever; default 120 s or OTT_COMM_TIMEOUT_MS)."""
        N.check(N.lib().ott_comm_set_timeout_ms(self._h, int(ms)))

    @staticmethod
    def unique_id() -> bytes:
        N.lib()
        N.preload_torch_rccl()
        buf = C.create_string_buffer(N.COMM_ID_BYTES)
        N.check(N.lib().ott_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, uid: bytes, rank: int, world: int, device: int) -> "Comm":
        if len(uid) != N.COMM_ID_BYTES:
            raise N.OttersError(f"RCCL unique id must be {N.COMM_ID_BYTES} bytes")
        N.lib()
        N.preload_torch_rccl()
        h = C.c_void_p()
        N.check(N.lib().ott_comm_create(C.c_char_p(uid), int(rank), int(world), int(device), C.byref(h)))
        return cls(h, rank, world)

    @classmethod
    def host(cls, rank: int, world: int, allgather: Callable[[bytes], bytes]) -> "Comm":
        def _cb(_user, send, recv, nbytes):
            try:
                out = allgather(C.string_at(send, nbytes))
                if len(out) != nbytes * world:
                    return 1
                C.memmove(recv, out, len(out))
                return 0
            except Exception:  # noqa: BLE001 -- must not unwind through the C frames; the library reports the failure
                return 1
        cb = N.ALLGATHER_FN(_cb)
        h = C.c_void_p()
        N.check(N.lib().ott_comm_create_host(int(rank), int(world), cb, None, C.byref(h)))
        return cls(h, rank, world, keep=cb)

    @classmethod
    def from_torch(cls, dist, device: int = 0, transport: str = "auto") -> "Comm":
        """`dist`: the torch.distributed module with an initialised default group.  transport "rccl": rank 0's RCCL id is
        broadcast over the group (control plane only) and every rank joins an RCCL communicator of its own; "host": the
        group itself carries the blocks (CPU tensors; e.g. gloo); "auto": rccl when the group's backend is nccl."""
        rank, world = dist.get_rank(), dist.get_world_size()
        if transport == "auto":
            transport = "rccl" if dist.get_backend() == "nccl" else "host"
        if transport == "rccl":
            import torch
            uid = cls.unique_id() if rank == 0 else bytes(N.COMM_ID_BYTES)
            if world > 1:  # the 128 id bytes as a plain uint8 tensor (no pickled objects on the wire)
                t = torch.frombuffer(bytearray(uid), dtype=torch.uint8)
                if dist.get_backend() == "nccl":
                    t = t.to(torch.device("cuda", device))
                dist.broadcast(t, src=0)
                uid = bytes(t.cpu().numpy().tobytes())
            return cls.rccl(uid, rank, world, device)
        import torch

        def allgather(b: bytes) -> bytes:
            if world == 1:
                return b
            t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
            out = torch.empty(world * t.numel(), dtype=torch.uint8)
            if dist.get_backend() == "nccl":  # a device-only backend: bounce through its GPU
                dev = torch.device("cuda", device)
                od = torch.empty(world * t.numel(), dtype=torch.uint8, device=dev)
                dist.all_gather_into_tensor(od, t.to(dev))
                out = od.cpu()
            else:
                dist.all_gather_into_tensor(out, t)
            return out.numpy().tobytes()
        return cls.host(rank, world, allgather)

    @property
    def transport(self) -> str:
        return N.lib().ott_comm_transport(self._h).decode()

    def all_gather_host(self, arr: np.ndarray) -> np.ndarray:
        """Equal-size host arrays of every rank, stacked in rank order (control data; also a barrier)."""
        a = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + a.shape, dtype=a.dtype)
        N.check(N.lib().ott_comm_all_gather_host(self._h, N.ptr(a), N.ptr(out), a.nbytes))
        return out

    def all_gather_bytes(self, payload: bytes):
        """Variable-size byte strings of every rank (sizes first, then the payloads padded to the longest)."""
        sizes = self.all_gather_host(np.array([len(payload)], dtype=np.int64)).ravel()
        longest = int(sizes.max())
        if longest == 0:
            return [b""] * self.world
        buf = np.zeros(longest, dtype=np.uint8)
        buf[: len(payload)] = np.frombuffer(payload, dtype=np.uint8)
        allb = self.all_gather_host(buf)
        return [allb[r, : int(sizes[r])].tobytes() for r in range(self.world)]

    def barrier(self) -> None:
        self.all_gather_host(np.zeros(1, dtype=np.int64))

    def info(self) -> dict:
        """What the transport itself reports (ott_comm_info): {"nranks": ncclCommCount, "version": ncclGetVersion}."""
        n, v = C.c_int(0), C.c_int(0)
        N.check(N.lib().ott_comm_info(self._h, C.byref(n), C.byref(v)))
        return {"nranks": n.value, "version": v.value}

    def close(self) -> None:
        if self._h is not None:
            self._fin.detach()
            N.lib().ott_comm_destroy(self._h)
            self._h = None

    # No __del__: ott_comm_destroy makes HIP and RCCL calls, and a __del__ that runs at interpreter teardown can find the
    # HIP runtime already gone.  weakref.finalize runs at collection time, or from atexit while the runtime is still up.


class ShardedPlan(VecQueryPlan):
    def __init__(self, sharded: "ShardedVecStore"):
        super().__init__()
        self._sharded = sharded

    def collect_arrays(self):
        rq = self.resolve()
        hits, counts = self._sharded._run(rq)
        if rq.mode != Mode.PerQuery:
            counts = np.bincount(hits["query"], minlength=rq.queries.shape[0]).tolist()
        return hits, counts


class ShardedVecStore:
    """This rank's shard + the comm.  `store.set_base_offset(base)` must hold the shard's first global row so hits carry
    global indices (src/meta_compute.rs:185); shards are in rank order.  `comm`: a `Comm`, or the torch.distributed
    module (then `Comm.from_torch` picks the transport: RCCL for an nccl group, host callback otherwise)."""

    def __init__(self, store: VecStore, comm, global_rows: Optional[int] = None):
        self.store = store
        self.comm = comm if isinstance(comm, Comm) else Comm.from_torch(comm, store.device)
        self.world, self.rank = self.comm.world, self.comm.rank
        self.dim = store.dim
        # VecStore::len of the WHOLE corpus: the shard sizes are exchanged once (a collective: every rank constructs its
        # ShardedVecStore at the same point), unless the caller states the total
        if global_rows is None:
            global_rows = int(self.comm.all_gather_host(np.array([store.len()], dtype=np.int64)).sum())
        self.global_rows = int(global_rows)

    def len(self) -> int:  # src/vec.rs:378
        return self.global_rows

    def query(self, queries, metric: Metric) -> ShardedPlan:
        plan = ShardedPlan(self)
        plan.with_query_vectors(queries).with_metric(metric)
        plan.vector_store = self  # resolve() only needs .dim and .len(): the default take is every row of the corpus
        return plan

    def _run(self, rq: ResolvedQuery, chunk_mask: Optional[np.ndarray] = None, use_device_row_mask: bool = False):
        """ott_query_sharded (collective).  Returns (hits, per-query counts); every rank gets the same."""
        nq = rq.queries.shape[0]
        perq = rq.mode == Mode.PerQuery
        store = self.store
        pool = self.global_rows if perq else self.global_rows * nq
        k_out = min(rq.k, pool)  # the whole job cannot return more than exists
        cap = max(k_out * (nq if perq else 1), 1)
        # ott_query_sharded's contract is cap >= k (or nq * k); a k beyond the corpus is clamped here, identically on every rank
        d = N.QueryDesc()
        d.queries = rq.queries.ctypes.data
        d.nq = nq
        d.metric, d.take, d.filter_cmp, d.filter_thr = rq.metric, rq.take, rq.filter_cmp, rq.filter_thr
        d.mode, d.k, d.path = rq.mode, k_out, rq.path
        keep = []
        if chunk_mask is not None:  # this shard's zonemap prune (bit c = local chunk c)
            cm = N.pack_bits(chunk_mask)
            keep.append(cm)
            d.chunk_mask = cm.ctypes.data
        if use_device_row_mask:
            d.use_device_row_mask = 1
        elif rq.row_mask is not None and rq.row_mask.size:
            rm = N.pack_bits(rq.row_mask)
            keep.append(rm)
            d.row_mask, d.row_mask_bits = rm.ctypes.data, int(rq.row_mask.size)
        out = np.empty(cap, dtype=N.HIT_DTYPE)
        n_out = C.c_uint64(0)
        per = (C.c_uint64 * nq)()
        st = N.Stats()
        N.check(N.lib().ott_query_sharded(store._handle(), self.comm._h, C.byref(d), N.ptr(out), cap, C.byref(n_out), per, C.byref(st)))
        store.last_stats = st.as_dict()
        return out[: n_out.value], [int(x) for x in per]


class ShardedMetaStore:
    """MetaStore sharded by contiguous chunk ranges: rank g holds the vectors AND the metadata columns of
    its rows (a MetaStore built from its slice).  A query prunes and masks locally (each shard's own zonemaps and
    HBM-resident columns), scores locally, and joins the other shards through the same single candidate exchange as
    ShardedVecStore; every rank ends up with the same MetaQueryResults (the column values of the k hits are
    materialised by the ranks that own them and exchanged over the comm)."""

    def __init__(self, meta, comm, base_row: int, global_rows: Optional[int] = None):
        self.meta = meta
        self.base = int(base_row)
        store = meta._store
        if store is None:  # a shard without rows still takes part in every collective
            store = VecStore(meta._dim if meta._dim else 1)
        store.set_base_offset(self.base)
        self.sharded = ShardedVecStore(store, comm, global_rows)
        self.comm = self.sharded.comm
        self._last_stats = None

    def query(self, query, metric: Metric) -> "ShardedMetaPlan":
        return ShardedMetaPlan(self, [np.ascontiguousarray(query, dtype=np.float32).ravel()], metric)

    def query_batch(self, queries, metric: Metric) -> "ShardedMetaPlan":
        return ShardedMetaPlan(self, [np.ascontiguousarray(q, dtype=np.float32).ravel() for q in queries], metric)

    def last_query_stats(self):
        return self._last_stats


class ShardedMetaPlan(MetaQueryPlan):
    def __init__(self, sms: ShardedMetaStore, queries, metric: Metric):
        super().__init__(sms.meta, queries, metric)
        self._sms = sms

    def collect(self):
        import time
        t0 = time.perf_counter()
        sms, st = self._sms, self._sms.meta
        if not self.queries:
            raise N.OttersError("No queries provided")
        rq, chunk_mask, compiled = self.resolve()  # this shard's zonemap prune (src/meta.rs:632-669)
        if self.take_count is None:
            rq.k = sms.sharded.global_rows  # default take = every row of the CORPUS (src/meta.rs:638-640), not of the shard
        prune = time.perf_counter() - t0
        with st._mask_lock:
            use_dev = False
            if compiled is not None and st._n_rows and st.row_mask_is_all_true(compiled, chunk_mask):
                pass  # zone statistics decide every row of every surviving chunk (MetaStore.row_mask_is_all_true)
            elif compiled is not None and st._n_rows:
                if st._device_mask_ok(compiled):
                    st.build_row_mask_device(compiled)
                    use_dev = True
                else:
                    rq.row_mask = st.build_row_mask_host(compiled)
            hits, _ = sms.sharded._run(rq, chunk_mask=chunk_mask if st._n_rows else None, use_device_row_mask=use_dev)
        g = sms.sharded.store.last_stats
        # stats of the whole job (src/meta.rs:711-720): sums over the shards
        evaluated = int(chunk_mask.sum()) if chunk_mask is not None else st._n_chunks
        mine = [st._n_chunks, st._n_chunks - evaluated, evaluated, int(g["vectors_compared"])]
        # materialise: each rank fills in the rows it owns (src/meta.rs:722-828), then the pieces are exchanged
        idx = hits["index"].astype(np.int64)
        own = (idx >= sms.base) & (idx < sms.base + st._n_rows)
        names = sorted(st._schema)
        local_rows = {int(i): {n: _cell(st._columns[n], int(i) - sms.base) for n in names} for i in idx[own]}
        # fixed schema, plain JSON (never pickle: with the host-callback transport the bytes come from whatever all-gather the
        # embedding host supplies): {"stats": [4 ints], "rows": [[global row, {column: cell or null}], ...]}
        payload = json.dumps({"stats": mine, "rows": [[i, cells] for i, cells in local_rows.items()]}, allow_nan=True).encode()
        parts = [_decode_cells(b, names) for b in sms.comm.all_gather_bytes(payload)]
        tot = np.sum([p[0] for p in parts], axis=0)
        rows = {}
        for p in parts:
            rows.update(p[1])
        from .col import Column
        data = {}
        for n in names:
            c = Column(n, st._schema[n])
            for i in idx:
                c.push(rows[int(i)][n])
            data[n] = c
        total = time.perf_counter() - t0
        sms._last_stats = MetaQueryStats(int(tot[0]), int(tot[1]), int(tot[2]), int(tot[3]), prune, max(total - prune, 0.0), 0.0, total,
                                         bytes_scanned=g["bytes_scanned"], path_used=g["path_used"], gpu_score_ms=g["score_ns"] / 1e6)
        return MetaQueryResults(names, data, [int(i) for i in idx], [float(x) for x in hits["score"]])


def _decode_cells(blob: bytes, names):
    """One rank's contribution to a materialisation exchange -> ([total, pruned, evaluated, compared], {row: {column: cell}}).
    Everything is checked against the fixed schema; a peer can hand over wrong VALUES, never code."""
    d = json.loads(blob.decode())
    stats = [int(x) for x in d["stats"]]
    if len(stats) != 4:
        raise N.OttersError("sharded materialisation: malformed stats block from a peer")
    rows = {}
    for i, cells in d["rows"]:
        if not isinstance(cells, dict) or sorted(cells) != list(names):
            raise N.OttersError("sharded materialisation: malformed row block from a peer")
        for v in cells.values():
            if not (v is None or isinstance(v, (int, float, str))):
                raise N.OttersError("sharded materialisation: unexpected cell type from a peer")
        rows[int(i)] = cells
    return stats, rows


def _cell(col, i: int):
    """Row i of a column as a Python value, None for NULL (what Column.push takes back)."""
    if col.null_mask()[i]:
        return None
    v = col.values()[i]
    return v if isinstance(v, str) else v.item()
