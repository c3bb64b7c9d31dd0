"""Ingest from HOST memory (src/meta.rs:203-281 builds a store from host vectors; src/vec.rs:357-376 add_vector/add_vectors):
ott_store_append = H2D copy of the rows + inverse norms on the GPU, measured against this box's own H2D peak (a plain
hipMemcpy of the same bytes from pinned and from pageable memory), as one flat buffer, as Vec<Vec<f32>>-shaped pieces, and as a
MetaStore build (2 metadata columns uploaded, zone statistics computed on the GPU).

    python benchmarks/ingest.py [rows] [dim]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Column, DataType, MetaStore, VecStore
from otters_amd import _native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
avail = 0
with open("/proc/meminfo") as f:
    for ln in f:
        if ln.startswith("MemAvailable:"):
            avail = int(ln.split()[1]) * 1024
if avail and n * dim * 4 > 0.45 * avail:  # the host copy of the corpus must leave the box plenty of room
    n = int(0.45 * avail / (dim * 4)) // 4096 * 4096
    print(f"(host memory available {avail / 1e9:.0f} GB: rows capped at {n})")
t0 = time.perf_counter()
rows = np.empty((n, dim), dtype=np.float32)
rng = np.random.default_rng(0)
for i in range(0, n, 500_000):  # (filled in pieces: one 30-GB uniform() call would double the host footprint)
    rows[i:i + 500_000] = rng.random((min(500_000, n - i), dim), dtype=np.float32) * 2 - 1
gb = rows.nbytes / 1e9
print(f"host rows {n} x {dim} = {gb:.2f} GB generated in {time.perf_counter() - t0:.1f} s\n")
print("| what | seconds | GB/s |")
print("|---|---|---|")


def line(what, dt):
    print(f"| {what} | {dt:.3f} | {gb / dt:.1f} |", flush=True)


# the box's own H2D rates for these bytes: hipMemcpy from pageable memory (what a Rust Vec<f32> is) and from pinned memory
N.lib()              # loads the HIP runtime this process uses (PyTorch's bundled one when torch is installed), RTLD_GLOBAL
hip = C.CDLL(None)   # ... and the plain-copy baseline calls the SAME runtime through the global namespace
dev = C.c_void_p()
piece = min(n, 2_000_000)
pb = piece * dim * 4
assert hip.hipMalloc(C.byref(dev), C.c_size_t(pb)) == 0
t = time.perf_counter()
for i in range(0, n - piece + 1, piece):
    assert hip.hipMemcpy(dev, C.c_void_p(rows[i:i + piece].ctypes.data), C.c_size_t(pb), 1) == 0
dt = time.perf_counter() - t
print(f"| plain hipMemcpy H2D, pageable source ({piece}-row pieces) | {dt:.3f} | {(n // piece) * pb / 1e9 / dt:.1f} |", flush=True)
pin = C.c_void_p()
assert hip.hipHostMalloc(C.byref(pin), C.c_size_t(pb), 0) == 0
C.memmove(pin, rows.ctypes.data, pb)
t = time.perf_counter()
for _ in range(4):
    assert hip.hipMemcpy(dev, pin, C.c_size_t(pb), 1) == 0
dt = time.perf_counter() - t
print(f"| plain hipMemcpy H2D, pinned source ({piece}-row pieces) | {dt / 4:.3f} | {pb / 1e9 / (dt / 4):.1f} |", flush=True)
hip.hipHostFree(pin)
hip.hipFree(dev)

for it in range(2):
    s = VecStore(dim)
    s.reserve(n)
    t = time.perf_counter()
    s.add_vectors(rows)
    line(f"ott_store_append, one flat buffer, reserved (run {it})", time.perf_counter() - t)
    s.close()
s = VecStore(dim)
t = time.perf_counter()
for i in range(0, n, 100_000):
    s.add_vectors(rows[i:i + 100_000])
line("ott_store_append in 100k-row pieces, no reserve (the store doubles as it grows)", time.perf_counter() - t)
s.close()
s = VecStore(dim)
s.reserve(n)
t = time.perf_counter()
for i in range(0, n, 1000):
    N.check(N.lib().ott_store_append(s._handle(), C.c_void_p(rows[i:i + 1000].ctypes.data), min(1000, n - i)))
line("ott_store_append in 1000-row pieces (add_vectors of small Vec<Vec<f32>> batches), reserved", time.perf_counter() - t)
s.close()
m = min(n, 200_000)
s = VecStore(dim)
s.reserve(m)
t = time.perf_counter()
for i in range(m):
    N.check(N.lib().ott_store_append(s._handle(), C.c_void_p(rows[i].ctypes.data), 1))
dt = time.perf_counter() - t
print(f"| ott_store_append ONE row per call (VecStore::add_vector), {m} rows | {dt:.3f} | {m * dim * 4 / 1e9 / dt:.3f} ({m / dt:.0f} rows/s) |", flush=True)
s.close()
# MetaStore build: vectors + 2 metadata columns to HBM, zone statistics on the GPU
chunk = np.arange(n) // 4096
cols = [Column.from_numpy("bucket", DataType.Int32, (chunk % 2).astype(np.int32)),
        Column.from_numpy("price", DataType.Float64, (chunk % 5) * 20.0 + np.random.default_rng(1).uniform(0, 25, n))]
t = time.perf_counter()
meta = MetaStore.from_columns(cols).with_vectors(rows).with_chunk_size(4096).build()
dt = time.perf_counter() - t
bs = meta.build_stats()
line(f"MetaStore build (chunk 4096, 2 columns; vectors {bs.vectors_ingest_duration:.3f} s, zonemaps {bs.zonemap_build_duration:.3f} s)", dt)
