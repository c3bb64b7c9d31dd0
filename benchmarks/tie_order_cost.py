"""What the reference's tie outcome costs when nothing is tied (the default of the Rust patch and the C++ mirror: tie_order 1 for
VecStore, 2 for MetaStore): one extra candidate per list (k + 1), the ambiguity check on the host, no second pass.  Headline
shape (10M x 768, one query, cosine top-10), config 2 (256 queries, top-100, merged), config 3's chunked form (chunk 4096,
every second chunk pruned, vec_filter 0.5 Gt) — wall per call through the Python mirror, median (min), tie_order 0 / 1 / 2
interleaved on ONE store.

    python benchmarks/tie_order_cost.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Cmp, Metric, VecStore

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = 768
s = VecStore(dim)
s.set_chunk_size(4096)
s.append_random(rows, 0x07735)
rng = np.random.default_rng(3)
q1 = rng.uniform(-1, 1, dim).astype(np.float32)
q256 = rng.uniform(-1, 1, (256, dim)).astype(np.float32)
n_chunks = (rows + 4095) // 4096
cmask = (np.arange(n_chunks) % 2) == 1


def call(kind):
    if kind == "headline":
        return s.query(q1, Metric.Cosine).take(10).collect_arrays()[0]
    if kind == "c2":
        return s.query(q256, Metric.Cosine).take(100).collect_arrays()[0]
    rq = s.query(q1, Metric.Cosine).filter(0.5, Cmp.Gt).take(10).resolve()
    return s._run(rq, chunk_mask=cmask)[0]


print(f"rows {rows} x {dim}; wall ms per call, median (min)")
print("| shape | tie_order 0 (canonical) | 1 (VecStore reference) | 2 (MetaStore reference) | results equal |")
print("|---|---|---|---|---|")
for kind, reps in (("headline", 40), ("c2", 9), ("c3", 40)):
    t = {0: [], 1: [], 2: []}
    res = {}
    for order in (0, 1, 2):
        s.set_option("tie_order", order)
        call(kind)
    for _ in range(reps):
        for order in (0, 1, 2):
            s.set_option("tie_order", order)
            t0 = time.perf_counter()
            res[order] = call(kind)
            t[order].append(time.perf_counter() - t0)
    same = all(np.array_equal(res[0]["index"], res[o]["index"]) and np.array_equal(res[0]["score"].view(np.uint32), res[o]["score"].view(np.uint32)) for o in (1, 2))
    cell = lambda o: f"{np.median(t[o]) * 1e3:.3f} ({np.min(t[o]) * 1e3:.3f})"
    print(f"| {kind} | {cell(0)} | {cell(1)} | {cell(2)} | {same} |", flush=True)
