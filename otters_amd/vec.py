"""Host-side mirror of otters' `vec` module (src/vec.rs) over the MI355X backend.

Same names, argument meaning and error strings as the reference:

    store = VecStore(3)
    store.add_vectors([[1, 0, 0], [0, 1, 0]])
    hits = store.query([1, 0, 0], Metric.Cosine).filter(0.5, Cmp.Gt).take(5).collect()

`collect()` raises `OttersError(msg)` where the reference returns `Err(msg)`.  The vectors
live in HBM behind libotters_hip.so; the scoring loop, filter and top-k run there
(ott_query).  There is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _native as N
from ._native import OttersError


class Metric(enum.IntEnum):  # src/vec.rs:11-16
    Cosine = 0
    Euclidean = 1
    DotProduct = 2


class TakeType(enum.IntEnum):  # src/vec.rs:18-22
    Min = 0
    Max = 1


class Cmp(enum.IntEnum):  # src/vec.rs:24-31 (0 is reserved for "no filter" in the ABI)
    Lt = 1
    Gt = 2
    Lte = 3
    Gte = 4
    Eq = 5


class Mode(enum.IntEnum):
    Merged = 0     # reference semantics: one top-k over all (query, row) pairs, src/vec.rs:217-219
    PerQuery = 1   # extension


class Path(enum.IntEnum):
    Auto = 0
    Exact = 1
    Mfma = 2


@dataclass(frozen=True)
class SearchResult:  # src/vec.rs:34-38
    index: int
    score: float

    def __str__(self) -> str:  # src/vec.rs:40-44
        return f"#{self.index} score={self.score:.6f}"


class QueryBatch:
    """src/vec.rs:320-336: a single vector or a batch of vectors."""

    def __init__(self, queries):
        self.queries: list = []
        if isinstance(queries, QueryBatch):
            self.queries = list(queries.queries)
            return
        if isinstance(queries, np.ndarray):
            if queries.ndim == 1:
                self.queries = [np.asarray(queries, dtype=np.float32)]
            else:  # a batch given as a matrix stays one (a sequence of row vectors): no per-row objects, no re-stacking
                self.queries = np.ascontiguousarray(queries, dtype=np.float32)
            return
        seq = list(queries)
        if len(seq) == 0:  # Vec<Vec<f32>>::new(): an empty batch
            self.queries = []
        elif isinstance(seq[0], (list, tuple, np.ndarray)):
            self.queries = [np.asarray(q, dtype=np.float32).ravel() for q in seq]
        else:
            self.queries = [np.asarray(seq, dtype=np.float32)]


@dataclass
class ResolvedQuery:
    """What `collect()` hands to ott_query once the plan is validated."""
    queries: np.ndarray            # [nq, dim] f32
    metric: int
    take: int
    k: int
    filter_cmp: int                # 0 = none
    filter_thr: float
    row_mask: Optional[np.ndarray]  # bool[<=n] or None
    mode: int = Mode.Merged
    path: int = Path.Auto


def infer_default_take_type(metric: Metric) -> TakeType:  # src/vec.rs:92-98
    return TakeType.Min if metric == Metric.Euclidean else TakeType.Max


class VecQueryPlan:
    """src/vec.rs:55-312: lazy builder; errors surface at collect()."""

    def __init__(self):  # VecQueryPlan::new, src/vec.rs:70-82
        self.query_vectors: Optional[list] = None
        self.search_metric: Optional[Metric] = None
        self.filter_criteria: Optional[tuple] = None
        self.take_type: Optional[TakeType] = None
        self.take_count: Optional[int] = None
        self.vector_store: Optional["VecStore"] = None
        self.error: Optional[str] = None
        self.row_mask: Optional[np.ndarray] = None
        self._mode = Mode.Merged
        self._path = Path.Auto

    @staticmethod
    def new() -> "VecQueryPlan":
        return VecQueryPlan()

    # -- builder -------------------------------------------------------------------------------
    def with_vector_store(self, store: "VecStore") -> "VecQueryPlan":  # src/vec.rs:118-121
        if self.error is None:
            self.vector_store = store
        return self

    def with_query_vectors(self, queries) -> "VecQueryPlan":  # src/vec.rs:123-138
        if self.error is None:
            self.query_vectors = QueryBatch(queries).queries
        return self

    def with_metric(self, metric: Metric) -> "VecQueryPlan":  # src/vec.rs:140-143
        if self.error is None:
            self.search_metric = Metric(metric)
        return self

    def with_row_mask(self, mask) -> "VecQueryPlan":  # src/vec.rs:145-148; bit i = row i, True = keep
        if self.error is None:
            self.row_mask = np.asarray(mask, dtype=bool).ravel()
        return self

    def filter(self, score: float, cmp: Cmp) -> "VecQueryPlan":  # src/vec.rs:150-153
        if self.error is None:
            self.filter_criteria = (float(np.float32(score)), Cmp(cmp))
        return self

    def _take_with_options(self, count: int, take_type: Optional[TakeType]) -> "VecQueryPlan":  # src/vec.rs:103-116
        if self.error is not None:
            return self
        self.take_count = int(count)
        if take_type is not None:
            self.take_type = take_type
        elif self.take_type is None and self.search_metric is not None:
            self.take_type = infer_default_take_type(self.search_metric)
        return self

    def take(self, count: int) -> "VecQueryPlan":  # src/vec.rs:155-158
        return self._take_with_options(count, None)

    def take_min(self, count: int) -> "VecQueryPlan":  # src/vec.rs:160-163
        return self._take_with_options(count, TakeType.Min)

    def take_max(self, count: int) -> "VecQueryPlan":  # src/vec.rs:165-168
        return self._take_with_options(count, TakeType.Max)

    # -- extensions (not in the reference) -----------------------------------------------------
    def per_query(self) -> "VecQueryPlan":
        """Return k hits for every query instead of one merged list (collect() then returns a list of lists)."""
        self._mode = Mode.PerQuery
        return self

    def with_path(self, path: Path) -> "VecQueryPlan":
        self._path = Path(path)
        return self

    # -- execution -----------------------------------------------------------------------------
    def validate(self) -> None:  # src/vec.rs:170-203
        if self.error is not None:
            raise OttersError(self.error)
        if self.query_vectors is None:
            raise OttersError("Query vectors or their norms are not set")
        if self.search_metric is None:
            raise OttersError("Search metric is not set")
        if self.vector_store is None:
            raise OttersError("Vector store is not set")
        if len(self.query_vectors) == 0:
            raise OttersError("No queries provided")
        dim = self.vector_store.dim
        if isinstance(self.query_vectors, np.ndarray):  # a matrix: every row has the same length
            if self.query_vectors.shape[1] != dim:
                raise OttersError(f"Query vector length {self.query_vectors.shape[1]} does not match expected dimension {dim}")
            return
        for q in self.query_vectors:
            if len(q) != dim:
                raise OttersError(f"Query vector length {len(q)} does not match expected dimension {dim}")

    def resolve(self) -> ResolvedQuery:
        """Validate and lower the plan (src/vec.rs:207-214).  Pure host logic, no GPU."""
        self.validate()
        store = self.vector_store
        if isinstance(self.query_vectors, np.ndarray):
            queries = self.query_vectors
        else:
            queries = np.ascontiguousarray(np.stack(self.query_vectors).astype(np.float32, copy=False))
        k = self.take_count if self.take_count is not None else store.len()  # src/vec.rs:213
        take = self.take_type if self.take_type is not None else TakeType.Max  # src/vec.rs:214
        fc, ft = (0, 0.0) if self.filter_criteria is None else (int(self.filter_criteria[1]), self.filter_criteria[0])
        return ResolvedQuery(queries=queries, metric=int(self.search_metric), take=int(take), k=max(int(k), 0),
                             filter_cmp=fc, filter_thr=ft, row_mask=self.row_mask, mode=int(self._mode), path=int(self._path))

    def collect(self):  # src/vec.rs:205-311
        hits, counts = self.collect_arrays()
        # .tolist() first: building the objects from Python scalars is several times faster than indexing a record array
        idx, sc = hits["index"].tolist(), hits["score"].tolist()
        if self._mode == Mode.PerQuery:
            out, o = [], 0
            for c in counts:
                out.append([SearchResult(i, x) for i, x in zip(idx[o:o + c], sc[o:o + c])])
                o += c
            return out
        return [SearchResult(i, x) for i, x in zip(idx, sc)]

    def collect_arrays(self):
        """collect() without the per-hit Python objects: (hits, counts) where `hits` is a NumPy record array
        (`index` u64, `score` f32, `query` u32 = which query of the batch scored it) sorted best-first -- per query,
        concatenated in query order, in per_query() mode -- and `counts[q]` is the number of hits of query q."""
        rq = self.resolve()
        store = self.vector_store
        hits, counts, stats = store._run(rq)
        store.last_stats = stats
        if rq.mode != Mode.PerQuery:  # merged: how many of the k hits each query of the batch contributed
            counts = np.bincount(hits["query"], minlength=rq.queries.shape[0]).tolist()
        return hits, counts


class VecStore:
    """src/vec.rs:338-412: row-major f32 vectors + per-row inverse norms, resident in HBM."""

    def __init__(self, dim: int, device: Optional[int] = None, devices: Optional[Sequence[int]] = None):  # VecStore::new, src/vec.rs:348-355
        """`devices`: a list of HIP device ordinals makes this ONE store over several GPUs of this process
        (ott_store_create_multi): one shard per entry, contiguous chunk ranges in row order; every method below and
        `.query(...).take(k).collect()` work unchanged and return the same bits as a single-GPU store.  An ordinal may
        repeat (several shards on one GPU).  Default, when the caller names NEITHER `device` nor `devices` and the process
        is not one rank of a multi-process job (WORLD_SIZE > 1: there every rank's shard is a single-GPU store on its own
        GPU): the environment variable OTTERS_HIP_DEVICES ("0,1,2,3"), else GPU 0.  A shard is brought in per 32768 rows (`set_option("multi_min_shard_rows", n)`; 0 = always split
        evenly): smaller stores stay on the first GPU and are answered by that shard's own query."""
        self.dim = int(dim)
        if (devices is None and device is None and os.environ.get("OTTERS_HIP_DEVICES")
                and int(os.environ.get("WORLD_SIZE", "1") or "1") <= 1):
            devices = [int(x) for x in os.environ["OTTERS_HIP_DEVICES"].split(",") if x.strip() != ""]
        self.devices = [int(x) for x in devices] if devices is not None else None
        self.device = int(self.devices[0]) if self.devices else int(device or 0)
        self._h = None        # ott_store*, created on first append
        self._n = 0
        self._chunk_size = None
        self._base_offset = 0
        self._reduce = None
        self._options: dict = {}
        self.last_stats: Optional[dict] = None
        # OTTERS_TIE_ORDER=reference: this mirror then returns, like the Rust patch and the C++ mirror do by default, the
        # reference's own outcome at exact score ties (one TopKCollector over the store); default here: the canonical order
        if os.environ.get("OTTERS_TIE_ORDER") == "reference":
            self._options["tie_order"] = 1

    @staticmethod
    def new(dim: int, device: Optional[int] = None, devices: Optional[Sequence[int]] = None) -> "VecStore":
        return VecStore(dim, device, devices)

    # -- native handle -------------------------------------------------------------------------
    def _handle(self):
        if self._h is None:
            h = C.c_void_p()
            if self.devices is not None:
                ids = (C.c_int * len(self.devices))(*self.devices)
                N.check(N.lib().ott_store_create_multi(self.dim, len(self.devices), ids, C.byref(h)))
            else:
                N.check(N.lib().ott_store_create(self.dim, self.device, C.byref(h)))
            self._h = h
            if self._chunk_size is not None:
                N.check(N.lib().ott_store_set_chunk_size(h, self._chunk_size))
            if self._base_offset:
                N.check(N.lib().ott_store_set_base_offset(h, self._base_offset))
            if self._reduce is not None:
                N.check(N.lib().ott_store_set_reduce_order(h, self._reduce))
            for name, value in self._options.items():
                N.check(N.lib().ott_store_set_option(h, name.encode(), value))
        return self._h

    def close(self) -> None:
        if self._h is not None:
            N.lib().ott_store_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ingest --------------------------------------------------------------------------------
    def add_vector(self, vector) -> None:  # src/vec.rs:357-371
        v = np.asarray(vector, dtype=np.float32).ravel()
        if v.size != self.dim:
            raise OttersError(f"Input vector length {v.size} does not match expected dimension {self.dim}")
        self._append(v[None, :])

    def add_vectors(self, vectors) -> None:  # src/vec.rs:373-376 (try_for_each: rows before a bad one stay added)
        if isinstance(vectors, np.ndarray) and vectors.ndim == 2:
            if vectors.shape[1] != self.dim:
                if vectors.shape[0]:
                    raise OttersError(f"Input vector length {vectors.shape[1]} does not match expected dimension {self.dim}")
                return
            self._append(vectors)
            return
        good = []
        for v in vectors:
            a = np.asarray(v, dtype=np.float32).ravel()
            if a.size != self.dim:
                if good:
                    self._append(np.stack(good))
                raise OttersError(f"Input vector length {a.size} does not match expected dimension {self.dim}")
            good.append(a)
        if good:
            self._append(np.stack(good))

    def _append(self, rows: np.ndarray) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.shape[0] == 0:
            return
        N.check(N.lib().ott_store_append(self._handle(), N.ptr(rows), rows.shape[0]))
        self._n += rows.shape[0]

    # extensions used by benchmarks / tests
    def reserve(self, n_rows: int) -> None:
        N.check(N.lib().ott_store_reserve(self._handle(), int(n_rows)))

    def append_random(self, n_rows: int, seed: int) -> None:
        """Synthetic uniform [-1,1) rows generated on the GPU (examples/demo.rs:4-7 distribution)."""
        N.check(N.lib().ott_store_append_random(self._handle(), int(n_rows), int(seed)))
        self._n += int(n_rows)

    def append_clustered(self, n_rows: int, seed: int, n_clusters: int, spread: float, aniso: float = 0.0) -> None:
        """Synthetic clustered rows generated on the GPU (ott_store_append_clustered): centres uniform [-1,1), members
        centre + spread * uniform noise, optionally shrinking along the dimensions (aniso)."""
        N.check(N.lib().ott_store_append_clustered(self._handle(), int(n_rows), int(seed), int(n_clusters), float(spread), float(aniso)))
        self._n += int(n_rows)

    def append_device(self, dev_ptr: int, n_rows: int) -> None:
        N.check(N.lib().ott_store_append_device(self._handle(), C.c_void_p(dev_ptr), int(n_rows)))
        self._n += int(n_rows)

    def write_rows(self, first_row: int, rows) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.dim)
        N.check(N.lib().ott_store_write_rows(self._handle(), int(first_row), N.ptr(rows), rows.shape[0]))

    def shards(self):
        """[(device, first_row, n_rows)] of the store's shards (one entry for a single-GPU store)."""
        h = self._handle()
        out = []
        for g in range(N.lib().ott_store_shard_count(h)):
            dev, first, cnt = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
            N.check(N.lib().ott_store_shard_info(h, g, C.byref(dev), C.byref(first), C.byref(cnt)))
            out.append((dev.value, first.value, cnt.value))
        return out

    def transport(self) -> str:
        """How a multi-GPU store's candidate blocks travel: "peer", "rccl", "undecided" (before the first query); "none" for one GPU."""
        return N.lib().ott_store_transport(self._handle()).decode()

    def set_chunk_size(self, chunk_size: int) -> None:
        self._chunk_size = max(int(chunk_size), 1)
        if self._h is not None:
            N.check(N.lib().ott_store_set_chunk_size(self._h, self._chunk_size))

    def set_base_offset(self, base: int) -> None:
        self._base_offset = int(base)
        if self._h is not None:
            N.check(N.lib().ott_store_set_base_offset(self._h, self._base_offset))

    def set_batch_image(self, enabled: bool) -> None:
        """Allow (default) or forbid the bf16 copies of the corpus the batch path keeps in HBM (the hi plane, half the size of
        the rows, built by the first batch query; the split image, the same size, built only when a batch needs the split
        pass).  Results never depend on them, only the speed of batches."""
        N.check(N.lib().ott_store_set_batch_image(self._handle(), 1 if enabled else 0))

    def prepare_batch(self) -> None:
        """Build the batch path's hi plane now (after loading / appending) rather than inside the first batch query."""
        if self._n:
            N.check(N.lib().ott_store_prepare_batch(self._handle()))

    def batch_ready(self) -> bool:
        """The batch path's hi plane exists and covers every row (built in the background after appends: option hi_prebuild)."""
        return bool(self._n) and bool(N.lib().ott_store_batch_ready(self._handle()))

    def set_option(self, name: str, value: int) -> None:
        """Behaviour switch of this store (ott_store_set_option; the sixteen names are listed in include/otters_hip.h:
        "tie_order", "hi_fmt", "hi_prebuild", ..., "force_fallback").  Results never depend on any but "tie_order"."""
        self._options[name] = int(value)
        if self._h is not None:
            N.check(N.lib().ott_store_set_option(self._h, name.encode(), int(value)))

    def set_tie_order(self, order: str) -> None:
        """Which of several EQUAL-scoring (row, query) pairs survives the cut at take(k).  "canonical" (default): the
        library's total order — better score, lower row, lower query.  "reference": what the reference's TopKCollector
        keeps (strict-improvement inserts in visit order — 8-row block, query, row — at the position its binary search
        returns, src/vec_compute.rs:236-277; one collector over the store, src/vec.rs:217-219).  Scores and every hit
        strictly better than the k-th score are the same either way."""
        self.set_option("tie_order", {"canonical": 0, "reference": 1, "reference_chunked": 2}[order])

    def set_reduce_order(self, order: int) -> None:
        self._reduce = int(order)
        if self._h is not None:
            N.check(N.lib().ott_store_set_reduce_order(self._h, self._reduce))

    def rows(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = self._n - first if n is None else n
        out = np.empty((n, self.dim), dtype=np.float32)
        if n:
            N.check(N.lib().ott_store_read_rows(self._handle(), int(first), int(n), N.ptr(out)))
        return out

    def inv_norms(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = self._n - first if n is None else n
        out = np.empty(n, dtype=np.float32)
        if n:
            N.check(N.lib().ott_store_read_inv_norms(self._handle(), int(first), int(n), N.ptr(out)))
        return out

    # -- reference API -------------------------------------------------------------------------
    def len(self) -> int:  # src/vec.rs:378-380
        return self._n

    def __len__(self) -> int:
        return self._n

    def is_empty(self) -> bool:  # src/vec.rs:382-384
        return self._n == 0

    def query(self, queries, metric: Metric) -> VecQueryPlan:  # src/vec.rs:386-411
        plan = VecQueryPlan()
        plan.query_vectors = QueryBatch(queries).queries
        plan.search_metric = Metric(metric)
        plan.vector_store = self
        return plan

    # -- execution -----------------------------------------------------------------------------
    def _run(self, rq: ResolvedQuery, chunk_mask: Optional[np.ndarray] = None, use_device_row_mask: bool = False):
        """ott_query.  Returns (hits HIT_DTYPE array, per-query counts, stats dict)."""
        nq = rq.queries.shape[0]
        if self._n == 0:  # nothing resident: VecQueryPlan::collect yields an empty Vec (src/vec.rs:222, 270)
            return np.zeros(0, dtype=N.HIT_DTYPE), [0] * nq, None
        perq = rq.mode == Mode.PerQuery
        pool = self._n if perq else self._n * nq
        k_eff = min(rq.k, pool)
        cap = max(k_eff * (nq if perq else 1), 1)
        out = np.empty(cap, dtype=N.HIT_DTYPE)  # ott_query writes n_out entries; only those are returned
        d = N.QueryDesc()
        d.queries = rq.queries.ctypes.data
        d.nq = nq
        d.metric, d.take, d.filter_cmp, d.filter_thr = rq.metric, rq.take, rq.filter_cmp, rq.filter_thr
        d.mode, d.k, d.path = rq.mode, rq.k, rq.path
        keep = []
        if chunk_mask is not None:
            cm = N.pack_bits(chunk_mask)
            keep.append(cm)
            d.chunk_mask = cm.ctypes.data
        if use_device_row_mask:
            d.use_device_row_mask = 1
        elif rq.row_mask is not None and rq.row_mask.size:
            rm = N.pack_bits(rq.row_mask)
            keep.append(rm)
            d.row_mask = rm.ctypes.data
            d.row_mask_bits = int(rq.row_mask.size)
        n_out = C.c_uint64(0)
        per = (C.c_uint64 * nq)()
        st = N.Stats()
        N.check(N.lib().ott_query(self._handle(), C.byref(d), N.ptr(out), cap, C.byref(n_out), per, C.byref(st)))
        hits = out if n_out.value == cap else out[: n_out.value].copy()
        return hits, list(per), st.as_dict()
