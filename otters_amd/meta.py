"""Host-side mirror of otters' `meta` module (src/meta.rs) over the MI355X backend.

    meta = MetaStore.from_columns([age, grade]).with_vectors(vectors).with_chunk_size(2).build()
    res = meta.query(q, Metric.Cosine).meta_filter(col("age").gt(15) & col("grade").eq("A")).vec_filter(0.5, Cmp.Gt).take(4).collect()

What stays on the host (as in a patched otters): builders, Expr::compile, the zonemap chunk
prune (build_chunk_mask_for_plan, src/meta.rs:407-544), string predicates, result
materialisation.  What runs on the GPU: the vectors (one contiguous HBM matrix; a chunk is a
row range), numeric/datetime row predicates (ott_store_eval_row_mask), scoring of the
surviving chunks, score filter, top-k and the merge (ott_query).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import math
import threading
import time
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from . import _native as N
from ._native import OttersError
from .col import Column, DataType, format_datetime
from .expr import CmpOp, ColumnFilter, CompiledFilter, Expr, ExprError
from .vec import Cmp, Metric, Mode, Path, ResolvedQuery, TakeType, VecStore

I32_MIN, I32_MAX = -(2 ** 31), 2 ** 31 - 1
I64_MIN, I64_MAX = -(2 ** 63), 2 ** 63 - 1


def _wrap_i32(v: int) -> int:  # Rust `i64 as i32`
    return ((int(v) + 2 ** 31) % 2 ** 32) - 2 ** 31


def _sat_i(v: float, lo: int, hi: int) -> int:  # Rust `f64 as i32/i64`: saturating, NaN -> 0
    if math.isnan(v):
        return 0
    if v <= lo:
        return lo
    if v >= hi:
        return hi
    return int(v)


class Bloom:
    """Per-chunk string zonemap (src/meta_compute.rs:99-116 uses the fastbloom crate, which is
    randomly seeded per build; false-positive patterns are therefore not part of the contract —
    only "no false negatives", which this deterministic double-hashing filter also guarantees)."""

    def __init__(self, n_bits: int, n_hashes: int):
        self.m = max(int(n_bits), 64)
        self.k = max(int(n_hashes), 1)
        self.bits = np.zeros((self.m + 63) // 64, dtype=np.uint64)

    @staticmethod
    def with_false_pos(p: float, expected: int) -> "Bloom":
        n = max(expected, 1)
        m = int(math.ceil(-n * math.log(p) / (math.log(2) ** 2)))
        return Bloom(m, round(m / n * math.log(2)))

    @staticmethod
    def with_num_bits(bits: int, expected: int) -> "Bloom":
        n = max(expected, 1)
        return Bloom(bits, round(bits / n * math.log(2)))

    @staticmethod
    def hash_pair(s: str):
        d = hashlib.blake2b(s.encode("utf-8"), digest_size=16).digest()
        return int.from_bytes(d[:8], "little"), int.from_bytes(d[8:], "little") | 1

    def _positions(self, s: str, hp=None):
        h1, h2 = hp if hp is not None else Bloom.hash_pair(s)
        return [(h1 + i * h2) % self.m for i in range(self.k)]

    def insert(self, s: str) -> None:
        for p in self._positions(s):
            self.bits[p >> 6] |= np.uint64(1) << np.uint64(p & 63)

    def contains(self, s: str) -> bool:
        return all((int(self.bits[p >> 6]) >> (p & 63)) & 1 for p in self._positions(s))


@dataclass
class MetaQueryStats:  # src/meta.rs:832-842 (durations in seconds) + device facts
    total_chunks: int
    pruned_chunks: int
    evaluated_chunks: int
    vectors_compared: int
    prune_duration: float
    score_duration: float
    merge_duration: float
    total_duration: float
    bytes_scanned: int = 0
    path_used: int = 0
    gpu_score_ms: float = 0.0

    def format(self) -> str:  # format_query_stats, display.rs:221-249
        rows = [("total_chunks", str(self.total_chunks)), ("pruned_chunks", str(self.pruned_chunks)),
                ("evaluated_chunks", str(self.evaluated_chunks)), ("vectors_compared", str(self.vectors_compared)),
                ("prune_ms", f"{self.prune_duration * 1e3:.3f}"), ("score_ms", f"{self.score_duration * 1e3:.3f}"),
                ("merge_ms", f"{self.merge_duration * 1e3:.3f}"), ("total_ms", f"{self.total_duration * 1e3:.3f}")]
        return ascii_table(["metric", "value"], rows, title="Last Meta Query Stats")


@dataclass
class MetaBuildStats:  # src/meta.rs:844-852
    n_rows: int
    dim: int
    n_chunks: int
    vectors_ingest_duration: float
    zonemap_build_duration: float
    build_total_duration: float

    def format(self) -> str:  # format_build_stats, display.rs:196-219
        rows = [("rows", str(self.n_rows)), ("dimensions", str(self.dim)), ("chunks", str(self.n_chunks)),
                ("vector_ingest_ms", f"{self.vectors_ingest_duration * 1e3:.3f}"),
                ("zonemap_build_ms", f"{self.zonemap_build_duration * 1e3:.3f}"),
                ("build_total_ms", f"{self.build_total_duration * 1e3:.3f}")]
        return ascii_table(["metric", "value"], rows, title="MetaStore Build Stats")


def ascii_table(headers, rows, title=None) -> str:  # AsciiTable::render, display.rs:32-96 (the title, when set, is the first line)
    if not headers:
        return ""
    w = [len(h) for h in headers]
    for r in rows:
        for i, c in enumerate(r):
            w[i] = max(w[i], len(c))
    sep = "+" + "+".join("-" * (x + 2) for x in w) + "+"
    line = lambda r: "|" + "|".join(" " + c.ljust(w[i]) + " " for i, c in enumerate(r)) + "|"
    return "\n".join(([title] if title is not None else []) + [sep, line(headers), sep] + [line(r) for r in rows] + [sep])


def _fmt_cell(c: Column, i: int) -> str:
    v = c.get(i)
    if v is None:
        return "NULL"
    dt = c.dtype()
    if dt in (DataType.Float32, DataType.Float64):
        return f"{v:.4f}"
    if dt == DataType.DateTime:
        return format_datetime(v)
    return str(v)


class MetaQueryResults:  # src/meta.rs:23-40
    def __init__(self, columns, data, indices, scores):
        self.columns: List[str] = columns
        self.data: Dict[str, Column] = data
        self.indices: List[int] = indices
        self.scores: List[float] = scores

    def len(self) -> int:
        return len(self.indices)

    def __len__(self) -> int:
        return len(self.indices)

    def is_empty(self) -> bool:
        return not self.indices

    def column(self, name: str) -> Optional[Column]:
        return self.data.get(name)

    def __str__(self) -> str:  # display.rs:164-188
        headers = ["index", "score"] + self.columns
        rows = [[str(ix), f"{sc:.6f}"] + [_fmt_cell(self.data[c], i) for c in self.columns]
                for i, (ix, sc) in enumerate(zip(self.indices, self.scores))]
        return ascii_table(headers, rows)  # (no title at this revision, display.rs:164-188; the README's sample, from an earlier one, shows "Query Results" above it)


class _Zone:
    """PackedRanges (src/meta.rs:71-76): SoA min / max / non_null per chunk, in the packed dtype."""

    def __init__(self, kind, mn, mx, non_null):
        self.kind, self.min, self.max, self.non_null = kind, mn, mx, non_null


def _build_numeric_zone(c: Column, chunk_size: int, n_chunks: int) -> _Zone:
    """build_zone_stat_for_range (src/meta_compute.rs:41-98, 117-130) for every chunk at once, then
    the packing of src/meta.rs:237-271 (f64 -> f32 narrowing, i64 -> i32 wrapping)."""
    dt = c.dtype()
    vals, nulls = c.values(), c.null_mask()
    n = vals.size
    pad = n_chunks * chunk_size - n
    live = ~nulls
    if dt in (DataType.Float32, DataType.Float64):
        v = vals.astype(np.float64)
        lo = np.where(live, v, np.inf)
        hi = np.where(live, v, -np.inf)
        lo = np.concatenate([lo, np.full(pad, np.inf)]).reshape(n_chunks, chunk_size)
        hi = np.concatenate([hi, np.full(pad, -np.inf)]).reshape(n_chunks, chunk_size)
        mn = np.fmin.reduce(lo, axis=1, initial=np.inf)  # f64::min ignores NaN
        mx = np.fmax.reduce(hi, axis=1, initial=-np.inf)
        if dt == DataType.Float32:
            with np.errstate(over="ignore"):
                mn, mx = mn.astype(np.float32), mx.astype(np.float32)
            kind = "f32"
        else:
            kind = "f64"
    else:
        v = vals.astype(np.int64)
        lo = np.where(live, v, I64_MAX)
        hi = np.where(live, v, I64_MIN)
        lo = np.concatenate([lo, np.full(pad, I64_MAX, dtype=np.int64)]).reshape(n_chunks, chunk_size)
        hi = np.concatenate([hi, np.full(pad, I64_MIN, dtype=np.int64)]).reshape(n_chunks, chunk_size)
        mn, mx = lo.min(axis=1), hi.max(axis=1)
        if dt == DataType.Int32:
            mn, mx = mn.astype(np.int32), mx.astype(np.int32)  # `as i32` wraps (all-null chunk: -1 / 0)
            kind = "i32"
        else:
            kind = "i64"
    nn = np.concatenate([live, np.zeros(pad, bool)]).reshape(n_chunks, chunk_size).sum(axis=1).astype(np.uint64)
    return _Zone(kind, mn, mx, nn)


def _range_sat(mn, mx, op: CmpOp, thr):
    """zonemap test, src/type_utils.rs:762-769"""
    if op == CmpOp.Eq:
        return (mn <= thr) & (thr <= mx)
    if op == CmpOp.Lt:
        return mn < thr
    if op == CmpOp.Lte:
        return mn <= thr
    if op == CmpOp.Gt:
        return mx > thr
    if op == CmpOp.Gte:
        return mx >= thr
    return np.ones(mn.shape, bool)  # Neq: always (still gated by non_null)


def _range_all(mn, mx, op: CmpOp, thr):
    """every value of a chunk with these bounds satisfies the ROW test (the universal counterpart of _range_sat)"""
    if op == CmpOp.Eq:
        return (mn == thr) & (mx == thr)
    if op == CmpOp.Neq:
        return (mx < thr) | (mn > thr)
    if op == CmpOp.Lt:
        return mx < thr
    if op == CmpOp.Lte:
        return mx <= thr
    if op == CmpOp.Gt:
        return mn > thr
    return mn >= thr


def _row_sat(v, op: CmpOp, thr):
    """row test, src/type_utils.rs:609-616"""
    with np.errstate(invalid="ignore"):
        if op == CmpOp.Eq:
            return v == thr
        if op == CmpOp.Neq:
            return v != thr
        if op == CmpOp.Lt:
            return v < thr
        if op == CmpOp.Lte:
            return v <= thr
        if op == CmpOp.Gt:
            return v > thr
        return v >= thr


class MetaStoreBuilder:  # src/meta.rs:62-306
    def __init__(self, schema: Dict[str, DataType], columns: Dict[str, Column], device: Optional[int] = None, devices=None):
        self.schema = schema
        self.columns = columns
        self.vectors = None
        self.chunk_size = 1024
        self.bloom = ("Fpr", 0.01)
        self.device = device
        self.devices = devices  # several GPUs of this process: one store over all of them (VecStore(devices=...))

    def with_vectors(self, vectors) -> "MetaStoreBuilder":
        self.vectors = vectors
        return self

    def with_chunk_size(self, chunk_size: int) -> "MetaStoreBuilder":  # src/meta.rs:86-89
        self.chunk_size = max(int(chunk_size), 1)
        return self

    def with_bloom_fpr(self, fpr: float) -> "MetaStoreBuilder":  # src/meta.rs:92-101
        f = min(max(fpr, 1e-2), 0.5) if math.isfinite(fpr) else 0.01
        self.bloom = ("Fpr", f)
        return self

    def with_bloom_bits(self, bits: int) -> "MetaStoreBuilder":  # src/meta.rs:106-110
        self.bloom = ("Bits", max(int(bits), 64))
        return self

    def with_column(self, name: str, column: Column) -> "MetaStoreBuilder":  # src/meta.rs:113-128
        if name not in self.schema:
            raise OttersError(f"unknown column '{name}' not present in schema")
        if self.schema[name] != column.dtype():
            raise OttersError(f"dtype mismatch for column '{name}': schema {self.schema[name].name}, got {column.dtype().name}")
        self.columns[name] = column
        return self

    def with_columns(self, columns) -> "MetaStoreBuilder":  # src/meta.rs:131-148
        for name, c in columns:
            self.with_column(name, c)
        return self

    def with_random_vectors(self, n_rows: int, dim: int, seed: int) -> "MetaStoreBuilder":
        """extension: rows generated on the GPU (benchmarks at sizes no host list could hold)"""
        self.vectors = ("random", int(n_rows), int(dim), int(seed))
        return self

    def build(self, _host_only: bool = False) -> "MetaStore":  # src/meta.rs:151-305
        if self.vectors is None:
            raise OttersError("vectors must be provided to build MetaStore")
        t0 = time.perf_counter()
        if isinstance(self.vectors, tuple) and self.vectors and self.vectors[0] == "random":
            _, n_rows, dim, seed = self.vectors
            mat = None
        else:
            seed = None
            if isinstance(self.vectors, np.ndarray) and self.vectors.ndim == 2:
                mat = np.ascontiguousarray(self.vectors, dtype=np.float32)
                n_rows, dim = mat.shape
            else:
                rows = [np.asarray(v, dtype=np.float32).ravel() for v in self.vectors]
                n_rows = len(rows)
                dim = rows[0].size if n_rows else 0
                for i, v in enumerate(rows):
                    if v.size != dim:
                        raise OttersError(f"vector at index {i} has dim {v.size}, expected {dim}")
                mat = np.stack(rows) if n_rows else np.zeros((0, 0), np.float32)
        for name in self.schema:
            if name not in self.columns:
                raise OttersError(f"missing column '{name}' in builder columns")
            c = self.columns[name]
            if c.len() != n_rows:
                raise OttersError(f"column '{name}' length {c.len()} does not match vectors length {n_rows}")
        if dim == 0 and n_rows > 0:
            raise OttersError("vector dimension cannot be zero")

        cs = self.chunk_size
        n_chunks = (n_rows + cs - 1) // cs
        t_ing = time.perf_counter()
        store = None
        if n_rows and not _host_only:
            store = VecStore(dim, self.device, self.devices)
            if store._options.get("tie_order") == 1:  # OTTERS_TIE_ORDER=reference: a MetaStore's outcome is one collector PER CHUNK
                store._options["tie_order"] = 2  # (any chunk size, src/meta.rs:86-89: 8-row blocks are counted from the chunk's first row)
            store.set_chunk_size(cs)
            store.reserve(n_rows)
            if mat is None:
                store.append_random(n_rows, seed)
            else:
                store.add_vectors(mat)
        ingest = time.perf_counter() - t_ing

        tz = time.perf_counter()
        zones: Dict[str, _Zone] = {}
        blooms: Dict[str, list] = {}
        str_nonnull: Dict[str, np.ndarray] = {}
        for name, dt in self.schema.items():
            c = self.columns[name]
            if n_chunks == 0:
                continue
            if dt == DataType.String:
                vals, nulls = c.values(), c.null_mask()
                bl, nn = [], np.zeros(n_chunks, dtype=np.uint64)
                for ch in range(n_chunks):
                    lo, hi = ch * cs, min((ch + 1) * cs, n_rows)
                    b = Bloom.with_false_pos(self.bloom[1], hi - lo) if self.bloom[0] == "Fpr" else Bloom.with_num_bits(self.bloom[1], hi - lo)
                    live = ~nulls[lo:hi]
                    for v in {v for v, ok in zip(vals[lo:hi], live) if ok}:  # inserting a value twice changes nothing
                        b.insert(v)
                    bl.append(b)
                    nn[ch] = int(live.sum())
                blooms[name], str_nonnull[name] = bl, nn
            elif store is None:
                zones[name] = _build_numeric_zone(c, cs, n_chunks)  # host numpy (CPU-only builds: tests)
        ms = MetaStore(dict(self.schema), dict(self.columns), cs, n_rows, dim, n_chunks, store, zones, blooms, str_nonnull)
        if store is not None and n_chunks:
            # numeric / datetime columns go to HBM once (row predicates run there) and their zonemaps are built there too
            for name, dt in self.schema.items():
                if dt != DataType.String:
                    zones[name] = ms.zone_stats_device(name)
        zdur = time.perf_counter() - tz
        ms._build_stats = MetaBuildStats(n_rows, dim, n_chunks, ingest, zdur, time.perf_counter() - t0)
        return ms


class MetaStore:  # src/meta.rs:48-60, 308-577
    def __init__(self, schema, columns, chunk_size, n_rows, dim, n_chunks, store, zones, blooms, str_nonnull):
        self._schema, self._columns = schema, columns
        self._chunk_size, self._n_rows, self._dim, self._n_chunks = chunk_size, n_rows, dim, n_chunks
        self._store: Optional[VecStore] = store
        self._zones, self._blooms, self._str_nonnull = zones, blooms, str_nonnull
        self._last_stats: Optional[MetaQueryStats] = None
        self._build_stats: Optional[MetaBuildStats] = None
        self._dev_cols: Dict[str, int] = {}
        self._str_codes: Dict[str, tuple] = {}
        self._mask_lock = threading.RLock()  # build_row_mask_device + the query that reads the mask: one critical section

    # -- constructors --------------------------------------------------------------------------------
    @staticmethod
    def from_columns(columns: List[Column], device: Optional[int] = None, devices=None) -> MetaStoreBuilder:  # src/meta.rs:332-347
        return MetaStoreBuilder({c.name(): c.dtype() for c in columns}, {c.name(): c for c in columns}, device, devices)

    @staticmethod
    def from_schema(schema, device: Optional[int] = None, devices=None) -> MetaStoreBuilder:  # src/meta.rs:350-364
        return MetaStoreBuilder({n: DataType(d) for n, d in schema}, {n: Column(n, DataType(d)) for n, d in schema}, device, devices)

    # -- accessors --------------------------------------------------------------------------------------
    def schema(self):
        return self._schema

    def columns(self):
        return self._columns

    def n_chunks(self) -> int:
        return self._n_chunks

    def chunk_size(self) -> int:
        return self._chunk_size

    def set_tie_order(self, order: str) -> None:
        """ "canonical" (default) or "reference": at exact score ties keep what the reference's MetaQueryPlan::collect keeps —
        one TopKCollector per surviving chunk (src/meta_compute.rs:153-192), the per-chunk lists concatenated in chunk order,
        sorted by score and truncated (src/meta.rs:699-709).  Any chunk size (a chunk's 8-row blocks are counted from its own first row)."""
        if self._store is not None:
            self._store.set_tie_order({"canonical": "canonical", "reference": "reference_chunked"}[order])

    def last_query_stats(self) -> Optional[MetaQueryStats]:
        return self._last_stats

    def build_stats(self) -> Optional[MetaBuildStats]:
        return self._build_stats

    def head_n(self, n: int) -> str:  # src/meta.rs:371-374 -> metastore_head, display.rs:125-161; printed and returned
        names = sorted(self._schema)
        rows = [[str(i)] + [_fmt_cell(self._columns[c], i) for c in names] for i in range(min(n, self._n_rows))]
        out = ascii_table(["index"] + names, rows,
                          title=f"MetaStore \u2022 rows={self._n_rows} \u2022 chunks={self._n_chunks} \u2022 chunk_size={self._chunk_size}")
        print(out)
        return out

    def head(self, n: int = 5) -> str:  # src/meta.rs:366-369
        return self.head_n(n)

    def print_build_stats(self) -> None:  # src/meta.rs:546-552
        print(self._build_stats.format() if self._build_stats else "(no build stats)")

    def print_last_query_stats(self) -> None:  # src/meta.rs:554-560
        print(self._last_stats.format() if self._last_stats else "(no query stats)")

    def print_last_stats(self) -> None:  # src/meta.rs:562-566
        self.print_build_stats()
        self.print_last_query_stats()

    # -- queries ----------------------------------------------------------------------------------------
    def query(self, query, metric: Metric) -> "MetaQueryPlan":  # src/meta.rs:569-571
        return MetaQueryPlan(self, [np.asarray(query, dtype=np.float32).ravel()], metric)

    def query_batch(self, queries, metric: Metric) -> "MetaQueryPlan":  # src/meta.rs:574-576
        return MetaQueryPlan(self, [np.asarray(q, dtype=np.float32).ravel() for q in queries], metric)

    # -- zonemap prune: build_chunk_mask_for_plan, src/meta.rs:407-428 ----------------------------------
    def build_chunk_mask_for_plan(self, compiled: CompiledFilter) -> np.ndarray:
        n = self._n_chunks
        if n == 0:
            return np.zeros(0, bool)
        acc = np.ones(n, bool)
        for clause in compiled.clauses:
            cm = np.zeros(n, bool)
            for leaf in clause:
                if leaf.kind == "Numeric":
                    cm |= self._numeric_leaf_chunk_mask(leaf)
                else:
                    cm |= self._string_leaf_chunk_mask(leaf)
            acc &= cm
        return acc

    def _numeric_leaf_chunk_mask(self, leaf: ColumnFilter) -> np.ndarray:  # src/meta.rs:431-521
        n = self._n_chunks
        z = self._zones.get(leaf.column)
        dt = self._schema.get(leaf.column)
        none = np.zeros(n, bool)
        if z is None:
            return none
        rhs = leaf.rhs
        if rhs.kind == "F64":
            if dt == DataType.Float32 and z.kind == "f32":
                with np.errstate(over="ignore"):
                    thr = np.float32(rhs.value)
            elif dt == DataType.Float64 and z.kind == "f64":
                thr = np.float64(rhs.value)
            elif z.kind == "i64":
                thr = np.int64(_sat_i(rhs.value, I64_MIN, I64_MAX))
            else:
                return none
        else:
            if dt == DataType.Int32 and z.kind == "i32":
                thr = np.int32(_wrap_i32(rhs.value))
            elif dt in (DataType.Int64, DataType.DateTime) and z.kind == "i64":
                thr = np.int64(rhs.value)
            else:
                return none
        return _range_sat(z.min, z.max, leaf.cmp, thr) & (z.non_null > 0)

    def _string_leaf_chunk_mask(self, leaf: ColumnFilter) -> np.ndarray:  # src/meta.rs:523-544
        n = self._n_chunks
        out = np.zeros(n, bool)
        bl = self._blooms.get(leaf.column)
        if bl is None:
            return np.ones(n, bool)  # conservatively keep when unknown
        nn = np.asarray(self._str_nonnull[leaf.column])
        if leaf.cmp == CmpOp.Neq:
            return nn > 0
        if leaf.cmp != CmpOp.Eq:
            return out
        # the literal is hashed ONCE and the filters of one shape (all chunks but a ragged last one) are tested together: their
        # words sit in one [chunks][words] matrix, built on first use (per chunk and query this loop was 3.3 us x 489 chunks)
        hp = Bloom.hash_pair(leaf.rhs)
        mats = self.__dict__.setdefault("_bloom_mats", {})
        if leaf.column not in mats:
            groups = {}
            for i, b in enumerate(bl):
                groups.setdefault((b.m, b.k), []).append(i)
            mats[leaf.column] = [(m, k, np.array(idx), np.stack([bl[i].bits for i in idx])) for (m, k), idx in groups.items()]
        for m, k, idx, words in mats[leaf.column]:
            keep = np.ones(idx.size, bool)
            for j in range(k):
                pos = (hp[0] + j * hp[1]) % m
                keep &= ((words[:, pos >> 6] >> np.uint64(pos & 63)) & np.uint64(1)).astype(bool)
            out[idx] = keep
        out &= nn > 0
        return out

    def row_mask_is_all_true(self, compiled: CompiledFilter, chunk_mask: np.ndarray) -> bool:
        """True when the zone statistics alone prove that build_row_mask_for_chunk (src/meta_compute.rs:194-232) would return
        an all-ones mask for EVERY surviving chunk: per chunk and clause some leaf on an integer / datetime column has no NULL
        in the chunk and bounds that satisfy the row test wholesale (row-level literal coercion).  The row mask can then be
        skipped — the result is the reference's, without evaluating 1 bit per row.  Float columns never qualify (their
        bounds ignore NaN rows, which fail every comparison but !=), nor do string leaves."""
        n = self._n_chunks
        if n == 0 or chunk_mask is None:
            return False
        lens = np.minimum(self._chunk_size, self._n_rows - np.arange(n, dtype=np.int64) * self._chunk_size)
        for clause in compiled.clauses:
            full = np.zeros(n, bool)
            for leaf in clause:
                z = self._zones.get(leaf.column)
                dt = self._schema.get(leaf.column)
                if leaf.kind != "Numeric" or z is None or z.kind not in ("i32", "i64") or dt not in (DataType.Int32, DataType.Int64, DataType.DateTime):
                    continue
                thr = self._row_literal(dt, leaf.rhs)
                full |= _range_all(z.min, z.max, leaf.cmp, thr) & (z.non_null.astype(np.int64) == lens)
            if not full[chunk_mask].all():
                return False
        return True

    # -- row masks: build_row_mask_for_chunk, src/meta_compute.rs:194-318 -----------------------------------
    @staticmethod
    def _row_literal(dt: DataType, rhs):
        """literal coercions of src/meta_compute.rs:249-283"""
        if dt == DataType.Float32:
            with np.errstate(over="ignore"):
                return np.float32(rhs.value)
        if dt == DataType.Float64:
            return np.float64(rhs.value)
        if dt == DataType.Int32:
            return np.int32(_wrap_i32(rhs.value) if rhs.kind == "I64" else _sat_i(rhs.value, I32_MIN, I32_MAX))
        return np.int64(rhs.value if rhs.kind == "I64" else _sat_i(rhs.value, I64_MIN, I64_MAX))

    def build_row_mask_host(self, compiled: CompiledFilter) -> np.ndarray:
        """All rows at once on the host: the checker of the GPU evaluator in the tests, and the path for plans the GPU
        evaluator does not take (a leaf naming an unknown column or one of the wrong kind)."""
        n = self._n_rows
        acc = np.ones(n, bool)
        for clause in compiled.clauses:
            cm = np.zeros(n, bool)
            for leaf in clause:
                c = self._columns.get(leaf.column)
                if c is None:
                    continue
                live = ~c.null_mask()
                if leaf.kind == "Numeric":
                    if c.dtype() == DataType.String:
                        continue
                    cm |= _row_sat(c.values(), leaf.cmp, self._row_literal(c.dtype(), leaf.rhs)) & live
                else:
                    vals = np.array(c.values(), dtype=object)
                    if leaf.cmp == CmpOp.Eq:
                        cm |= (vals == leaf.rhs) & live
                    elif leaf.cmp == CmpOp.Neq:
                        cm |= (vals != leaf.rhs) & live
            acc &= cm
        return acc

    def _string_codes(self, name: str):
        """Dictionary encoding of a string column, built once: ({value: code}, int32 codes).  The reference compares the
        strings row by row inside every surviving chunk (src/meta_compute.rs:291-318); `==` / `!=` on the strings is
        `==` / `!=` on the codes, so the row predicate runs on the GPU over a resident Int32 column like a numeric one."""
        if name not in self._str_codes:
            c = self._columns[name]
            vals = np.asarray(c.values(), dtype=object)
            uniq, inv = np.unique(vals.astype(str), return_inverse=True) if vals.size else (np.zeros(0, str), np.zeros(0, np.int64))
            self._str_codes[name] = ({u: i for i, u in enumerate(uniq.tolist())}, inv.astype(np.int32))
        return self._str_codes[name]

    def _device_column(self, name: str) -> int:
        if name not in self._dev_cols:
            c = self._columns[name]
            if c.dtype() == DataType.String:
                vals, dt = np.ascontiguousarray(self._string_codes(name)[1]), int(DataType.Int32)
            else:
                vals, dt = np.ascontiguousarray(c.values()), int(c.dtype())
            nulls = N.pack_bits(c.null_mask()) if c.null_mask().any() else None
            cid = C.c_uint32(0)
            N.check(N.lib().ott_store_add_column(self._store._handle(), dt, N.ptr(vals), N.ptr(nulls), vals.size, C.byref(cid)))
            self._dev_cols[name] = cid.value
        return self._dev_cols[name]

    def zone_stats_device(self, name: str):
        """build_zone_stat_for_range for every chunk on the GPU (src/meta_compute.rs:41-98, 117-130), packed like
        src/meta.rs:237-271.  Returns a _Zone equal to the host-built one."""
        c = self._columns[name]
        dt = c.dtype()
        cid = self._device_column(name)
        is_f = dt in (DataType.Float32, DataType.Float64)
        mn = np.zeros(self._n_chunks, dtype=np.float64 if is_f else np.int64)
        mx = np.zeros_like(mn)
        nn = np.zeros(self._n_chunks, dtype=np.uint64)
        N.check(N.lib().ott_store_zone_stats(self._store._handle(), cid, self._chunk_size, N.ptr(mn), N.ptr(mx), N.ptr(nn)))
        if dt == DataType.Float32:
            with np.errstate(over="ignore"):
                return _Zone("f32", mn.astype(np.float32), mx.astype(np.float32), nn)
        if dt == DataType.Float64:
            return _Zone("f64", mn, mx, nn)
        if dt == DataType.Int32:
            return _Zone("i32", mn.astype(np.int32), mx.astype(np.int32), nn)
        return _Zone("i64", mn, mx, nn)

    def _device_mask_ok(self, compiled: CompiledFilter) -> bool:
        """Every leaf names a known column of a kind the GPU evaluator takes (a Numeric leaf on a String column or
        an unknown column matches nothing in the host builder: those plans stay there)."""
        for cl in compiled.clauses:
            for leaf in cl:
                c = self._columns.get(leaf.column)
                if c is None or (leaf.kind == "Numeric") == (c.dtype() == DataType.String):
                    return False
        return True

    def build_row_mask_device(self, compiled: CompiledFilter, fetch: bool = False):
        """CNF evaluated on the GPU over HBM-resident columns (numeric, datetime, dictionary-coded strings)."""
        leaves = []
        for ci, clause in enumerate(compiled.clauses):
            for leaf in clause:
                c = self._columns[leaf.column]
                lf = N.Leaf()
                lf.column, lf.op, lf.clause = self._device_column(leaf.column), int(leaf.cmp), ci
                if leaf.kind == "String":
                    # src/meta_compute.rs:291-318: Eq / Neq on the non-null rows, every other operator matches nothing.
                    # A literal absent from the dictionary gets code -1 (no row has it); "matches nothing" is Eq -1.
                    code = self._string_codes(leaf.column)[0].get(leaf.rhs, -1)
                    if leaf.cmp not in (CmpOp.Eq, CmpOp.Neq):
                        lf.op, code = int(CmpOp.Eq), -1
                    lf.lit_i64 = int(code)
                elif c.dtype() in (DataType.Float32, DataType.Float64):
                    lf.lit_f64 = float(self._row_literal(c.dtype(), leaf.rhs))
                else:
                    lf.lit_i64 = int(self._row_literal(c.dtype(), leaf.rhs))
                leaves.append(lf)
            if not clause:  # an empty clause is an OR over nothing: no row passes
                raise OttersError("empty clause in compiled filter")
        arr = (N.Leaf * max(len(leaves), 1))(*leaves)
        out = np.zeros(max((self._n_rows + 63) // 64, 1), dtype=np.uint64) if fetch else None
        N.check(N.lib().ott_store_eval_row_mask(self._store._handle(), arr, len(leaves), len(compiled.clauses), N.ptr(out)))
        return N.unpack_bits(out, self._n_rows) if fetch else None


class MetaQueryPlan:  # src/meta.rs:579-830
    def __init__(self, store: MetaStore, queries, metric: Metric):
        self.store = store
        self.queries = queries
        self.metric = Metric(metric)
        self._meta_filter: Optional[CompiledFilter] = None
        self.meta_error: Optional[str] = None
        self._vec_filter = None
        self.take_type: Optional[TakeType] = None
        self.take_count: Optional[int] = None
        self._path = Path.Auto

    def meta_filter(self, expr: Expr) -> "MetaQueryPlan":  # src/meta.rs:605-616 (error deferred to collect)
        try:
            self._meta_filter = expr.compile(self.store._schema)
            self.meta_error = None
        except ExprError as e:
            self.meta_error = f"meta_filter compile error: {e}"
        return self

    def vec_filter(self, score: float, cmp: Cmp) -> "MetaQueryPlan":  # src/meta.rs:618-621
        self._vec_filter = (float(np.float32(score)), Cmp(cmp))
        return self

    def take(self, k: int) -> "MetaQueryPlan":  # src/meta.rs:623-630
        self.take_count = int(k)
        self.take_type = TakeType.Min if self.metric == Metric.Euclidean else TakeType.Max
        return self

    def with_path(self, path: Path) -> "MetaQueryPlan":
        self._path = Path(path)
        return self

    def resolve(self):
        """Host-side part of collect (src/meta.rs:632-669): k / take defaults, zonemap prune.
        Returns (ResolvedQuery, chunk_mask or None, compiled filter or None)."""
        if self.meta_error is not None:
            raise OttersError(self.meta_error)
        st = self.store
        k = self.take_count if self.take_count is not None else st._n_rows  # src/meta.rs:638-640
        take = self.take_type if self.take_type is not None else (TakeType.Min if self.metric == Metric.Euclidean else TakeType.Max)
        for q in self.queries:
            if st._n_rows and q.size != st._dim:
                raise OttersError(f"Query vector length {q.size} does not match expected dimension {st._dim}")
        chunk_mask = st.build_chunk_mask_for_plan(self._meta_filter) if self._meta_filter is not None else None
        fc, ft = (0, 0.0) if self._vec_filter is None else (int(self._vec_filter[1]), self._vec_filter[0])
        q = np.ascontiguousarray(np.stack(self.queries)) if self.queries else np.zeros((0, st._dim), np.float32)
        rq = ResolvedQuery(queries=q, metric=int(self.metric), take=int(take), k=max(int(k), 0), filter_cmp=fc, filter_thr=ft,
                           row_mask=None, mode=int(Mode.Merged), path=int(self._path))
        return rq, chunk_mask, self._meta_filter

    def collect(self) -> MetaQueryResults:  # src/meta.rs:632-829
        t0 = time.perf_counter()
        rq, chunk_mask, compiled = self.resolve()
        st = self.store
        prune = time.perf_counter() - t0
        total_chunks = st._n_chunks
        hits = np.zeros(0, dtype=N.HIT_DTYPE)
        gstats = None
        if not self.queries:
            raise OttersError("No queries provided")
        if st._store is not None and st._n_rows and (chunk_mask is None or chunk_mask.any()):
            use_dev = False
            # the device row mask is store-global state (ott_store_eval_row_mask writes it, the query reads it): building it and
            # querying with it are one critical section per MetaStore, so two threads filtering one store cannot score with
            # each other's mask (the reference's MetaStore is !Sync, src/meta.rs:54: there the compiler forbids the race)
            with st._mask_lock:
                if compiled is not None and st.row_mask_is_all_true(compiled, chunk_mask):
                    pass  # the zone statistics already decide every row of every surviving chunk: no row mask needed
                elif compiled is not None:
                    if st._device_mask_ok(compiled):
                        st.build_row_mask_device(compiled)  # numeric, datetime and (dictionary-coded) string leaves alike
                        use_dev = True
                    else:
                        rq.row_mask = st.build_row_mask_host(compiled)
                hits, _, gstats = st._store._run(rq, chunk_mask=chunk_mask, use_device_row_mask=use_dev)
            st._store.last_stats = gstats
        evaluated = int(chunk_mask.sum()) if chunk_mask is not None else total_chunks
        if gstats is not None:
            compared, score_d, merge_d = gstats["vectors_compared"], gstats["score_ns"] / 1e9, gstats["merge_ns"] / 1e9
        else:
            cs, n = st._chunk_size, st._n_rows
            lens = np.minimum(cs, n - np.arange(total_chunks) * cs) if total_chunks else np.zeros(0, int)
            compared = int((lens[chunk_mask] if chunk_mask is not None else lens).sum()) * len(self.queries)
            score_d = merge_d = 0.0
        total = time.perf_counter() - t0
        st._last_stats = MetaQueryStats(total_chunks, total_chunks - evaluated, evaluated, int(compared), prune,
                                        max(total - prune - merge_d, 0.0), merge_d, total,
                                        bytes_scanned=gstats["bytes_scanned"] if gstats else 0,
                                        path_used=gstats["path_used"] if gstats else 0, gpu_score_ms=score_d * 1e3)
        indices = [int(i) for i in hits["index"]]
        scores = [float(s) for s in hits["score"]]
        names = sorted(st._schema)  # src/meta.rs:723-724
        data = {name: st._columns[name].take(indices) for name in names}  # src/meta.rs:728-821
        return MetaQueryResults(names, data, indices, scores)
