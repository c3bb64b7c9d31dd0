"""Multi-GPU: the corpus sharded by contiguous chunk ranges, one process per GPU.

The reference fans chunks out over a rayon pool and then concat-sort-truncates the per-chunk
top-k lists (src/meta.rs:678-709).  Here rank g owns rows [base_g, base_g + n_g) of the global
corpus in its own HBM; a query is scored on every shard independently (no data-path
collective), then ONE exchange — an all-gather of the fixed-size per-GPU top-k candidate
lists over RCCL/xGMI (k x 16 B per GPU: latency-bound, far below link bandwidth) — and the
same final merge kernel on every rank.  `torch.distributed` is plumbing only: it moves the
candidate bytes; scoring, top-k and the merge are libotters_hip kernels.

The exchange and the host-side reference merge are plain functions (`pack_candidates`,
`gather_candidates`, `merge_candidates_host`) so the world_size>1 logic is covered on CPU with
the gloo backend; the product path always scores and merges on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

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
    """This rank's shard + the process group.  `store.set_base_offset(base)` must hold the shard's
    first global row so hits carry global indices (src/meta_compute.rs:185)."""

    def __init__(self, store: VecStore, dist, global_rows: Optional[int] = None):
        self.store = store
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.global_rows = global_rows
        self.dim = store.dim
        self._gather_buf = None
        self._local_buf = None
        self._cnt_buf = None

    def len(self) -> int:
        return self.global_rows if self.global_rows is not None else self.store.len() * self.world

    def query(self, queries, metric: Metric) -> ShardedPlan:
        plan = ShardedPlan(self)
        plan.with_query_vectors(queries).with_metric(metric)
        plan.vector_store = self  # resolve() only needs .dim and .len()
        return plan

    def _run(self, rq: ResolvedQuery, chunk_mask: Optional[np.ndarray] = None, use_device_row_mask: bool = False):
        import torch
        nq = rq.queries.shape[0]
        perq = rq.mode == Mode.PerQuery
        if rq.k > 512:
            raise N.OttersError("sharded queries support take(k) with k <= 512")
        store = self.store
        # slots per candidate list: a shard cannot contribute more than it holds; every rank must agree on the
        # block size, so it is derived from k and the (equal) nominal shard size only
        cap = int(min(max(rq.k, 1), 512))
        groups = nq if perq else 1
        block = groups * cap * 16
        dev = torch.device("cuda", store.device)
        if self._local_buf is None or self._local_buf.numel() != block:
            self._local_buf = torch.empty(block, dtype=torch.uint8, device=dev)
            self._gather_buf = torch.empty(self.world * block, dtype=torch.uint8, device=dev)
            self._cnt_buf = torch.zeros(1, dtype=torch.int64, device=dev)
        d = N.QueryDesc()
        d.queries = rq.queries.ctypes.data
        d.nq = nq
        d.metric, d.take, d.filter_cmp, d.filter_thr = rq.metric, rq.take, rq.filter_cmp, rq.filter_thr
        d.mode, d.k, d.path = rq.mode, min(rq.k, cap), rq.path
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
        st = N.Stats()
        # score this shard; the k best stay in HBM (sentinel padded)
        N.check(N.lib().ott_query_device(store._handle(), C.byref(d), C.c_void_p(self._local_buf.data_ptr()), groups * cap,
                                         C.c_void_p(self._cnt_buf.data_ptr()), C.byref(st)))
        # (ott_query_device returns once its stream has drained: the block is ready for the collective's stream)
        store.last_stats = st.as_dict()
        # the one exchange: all-gather of fixed-size candidate blocks (RCCL over xGMI)
        if self.dist.get_backend() == "nccl":
            self.dist.all_gather_into_tensor(self._gather_buf, self._local_buf)
            torch.cuda.current_stream(dev).synchronize()
        else:  # e.g. gloo: stage the k*16-byte blocks through the host
            host = gather_candidates(self.dist, self._local_buf.cpu())
            self._gather_buf.copy_(host)
            torch.cuda.current_stream(dev).synchronize()
        out = np.zeros(groups * cap, dtype=N.HIT_DTYPE)
        n_out = C.c_uint64(0)
        per = (C.c_uint64 * groups)()
        N.check(N.lib().ott_merge_hits_device_grouped(store._handle(), C.c_void_p(self._gather_buf.data_ptr()), self.world, groups, cap,
                                                      rq.take, min(rq.k, cap), N.ptr(out), C.byref(n_out), per))
        return out[: n_out.value], [int(x) for x in per]


class ShardedMetaStore:
    """MetaStore sharded by contiguous chunk ranges: rank g holds the vectors AND the metadata columns of
    its rows (a MetaStore built from its slice).  A query prunes and masks locally (each shard's own zonemaps and
    HBM-resident columns), scores locally, and joins the other shards through the same single candidate exchange as
    ShardedVecStore; every rank ends up with the same MetaQueryResults (the column values of the k hits are
    materialised by the ranks that own them and exchanged as Python objects: k <= 512 rows)."""

    def __init__(self, meta, dist, base_row: int, global_rows: Optional[int] = None):
        self.meta = meta
        self.dist = dist
        self.base = int(base_row)
        if meta._store is not None:
            meta._store.set_base_offset(self.base)
        self.sharded = ShardedVecStore(meta._store, dist, global_rows) if meta._store is not None else None
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
        prune = time.perf_counter() - t0
        use_dev = False
        if compiled is not None:
            if st._device_mask_ok(compiled):
                st.build_row_mask_device(compiled)
                use_dev = True
            else:
                rq.row_mask = st.build_row_mask_host(compiled)
        rq.k = min(rq.k, 512) if self.take_count is None else rq.k  # default take = every row: capped to what a shard exchange carries
        hits, _ = sms.sharded._run(rq, chunk_mask=chunk_mask, use_device_row_mask=use_dev)
        g = sms.sharded.store.last_stats
        # stats of the whole job (src/meta.rs:711-720): sums over the shards
        evaluated = int(chunk_mask.sum()) if chunk_mask is not None else st._n_chunks
        mine = np.array([st._n_chunks, st._n_chunks - evaluated, evaluated, g["vectors_compared"]], dtype=np.int64)
        parts = [None] * sms.dist.get_world_size()
        # materialise: each rank fills in the rows it owns (src/meta.rs:722-828), then the pieces are exchanged
        idx = hits["index"].astype(np.int64)
        own = (idx >= sms.base) & (idx < sms.base + st._n_rows)
        names = sorted(st._schema)
        local_rows = {int(i): {n: _cell(st._columns[n], int(i) - sms.base) for n in names} for i in idx[own]}
        sms.dist.all_gather_object(parts, (mine.tolist(), local_rows))
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


def _cell(col, i: int):
    """Row i of a column as a Python value, None for NULL (what Column.push takes back)."""
    if col.null_mask()[i]:
        return None
    v = col.values()[i]
    return v if isinstance(v, str) else v.item()
