"""Ingest throughput: ott_store_append from a host buffer (H2D copy + inverse norms on the GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import VecStore

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rows = np.random.default_rng(0).uniform(-1, 1, (n, dim)).astype(np.float32)
for it in range(3):
    s = VecStore(dim)
    s.reserve(n)
    t = time.perf_counter()
    s.add_vectors(rows)
    dt = time.perf_counter() - t
    print(f"append {n} x {dim} ({rows.nbytes / 1e9:.2f} GB): {dt * 1e3:.1f} ms = {rows.nbytes / dt / 1e9:.1f} GB/s", flush=True)
    s.close()
s = VecStore(dim)
t = time.perf_counter()
for i in range(0, n, 100_000):
    s.add_vectors(rows[i:i + 100_000])
dt = time.perf_counter() - t
print(f"append in 100k-row pieces without reserve: {dt * 1e3:.1f} ms = {rows.nbytes / dt / 1e9:.1f} GB/s")
