#!/usr/bin/env python3
"""demo.py — the reference's examples/demo.rs on the MI355X backend (BASELINE config 0 plumbing).

    python examples/demo.py [n_size=1000] [dim=100]
    OTTERS_HIP_DEVICES=0,1,2,3 python examples/demo.py 1000000 256      # the same store over four GPUs of this process

Builds a MetaStore whose metadata is hand-tuned per 128-row chunk so that even chunks prune
(examples/demo.rs:36-77), runs the same cosine + meta_filter + vec_filter(0.1, Gt) + take(5)
query (examples/demo.rs:105-113) and prints the tables and stats."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Cmp, Column, DataType, MetaStore, Metric, col  # noqa: E402


def main() -> None:
    n_size = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rng = np.random.default_rng()
    chunk = 128
    g = np.arange(n_size) // chunk
    even = g % 2 == 0
    columns = [
        Column("name", DataType.String).from_([f"item_{i}" for i in range(n_size)]),
        Column.from_numpy("price", DataType.Float64, np.where(even, 80.0, 10.0) + (np.arange(n_size) % 20)),
        Column("mfg", DataType.DateTime).from_(["2024-01-01" if e else "2024-07-01" for e in even]),
        Column("exp", DataType.DateTime).from_(["2024-12-31" if e else "2025-12-31" for e in even]),
        Column.from_numpy("version", DataType.Int32, np.where(even, 1, 3)),
    ]
    vectors = rng.uniform(-1, 1, (n_size, dim)).astype(np.float32)
    meta = MetaStore.from_columns(columns).with_vectors(vectors).with_chunk_size(chunk).build()
    print("=== MetaStore built ===")
    b = meta.build_stats()
    print(f"rows={b.n_rows} dim={b.dim} chunks={b.n_chunks} ingest={b.vectors_ingest_duration * 1e3:.3f} ms "
          f"zonemaps={b.zonemap_build_duration * 1e3:.3f} ms total={b.build_total_duration * 1e3:.3f} ms")
    print("\n=== MetaStore Head (ASCII table) ===")
    meta.head()
    res = (meta.query(rng.uniform(-1, 1, dim).astype(np.float32), Metric.Cosine)
           .meta_filter(col("price").lt(50.0) & col("version").gte(2) & col("exp").gte("2025-01-01"))
           .vec_filter(0.1, Cmp.Gt).take(5).collect())
    print("\n=== Meta query top 5 (ASCII table) ===")
    print(res)
    meta.print_last_query_stats()
    print("\n=== Access result columns (head) ===")
    for name in ("name", "price", "version"):
        c = res.column(name)
        if c is not None:
            c.head()


if __name__ == "__main__":
    main()
