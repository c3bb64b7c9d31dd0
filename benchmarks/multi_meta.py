"""MetaStore on the in-process multi-GPU store against one store (all shards on GPU 0): build, a zonemap-pruned query with a
device row mask, a query whose string leaf needs the host mask."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Cmp, Column, DataType, MetaStore, Metric, col
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dim, cs = 128, 4096
vec = np.random.default_rng(0).uniform(-1, 1, (n, dim)).astype(np.float32)
chunk = np.arange(n) // cs
def cols():
    r = np.random.default_rng(1)
    return [Column.from_numpy("bucket", DataType.Int32, (chunk % 2).astype(np.int32)),
            Column.from_numpy("price", DataType.Float64, (chunk % 5) * 20.0 + r.uniform(0, 25, n)),
            Column.from_numpy("grade", DataType.String, np.array(["A", "B", "C", "D"])[(chunk + r.integers(0, 2, n)) % 4])]
q = np.random.default_rng(2).uniform(-1, 1, dim).astype(np.float32)
print(f"| {n} x {dim}, chunk {cs} | shards | build s | pruned query (device mask) us | string-leaf query (host mask) us | no filter us |")
print("|---|---|---|---|---|---|")
for devs in (None, [0] * 4, [0] * 8):
    t = time.perf_counter()
    m = MetaStore.from_columns(cols(), devices=devs).with_vectors(vec).with_chunk_size(cs).build()
    tb = time.perf_counter() - t
    res = []
    for f in (lambda: col("bucket").eq(1) & col("price").lt(60.0), lambda: col("grade").eq("A") & col("price").lt(60.0), None):
        ts = []
        for it in range(40):
            p = m.query(q, Metric.Cosine)
            if f is not None: p = p.meta_filter(f())
            t = time.perf_counter(); r = p.take(10).collect(); ts.append(time.perf_counter() - t)
        res.append(np.median(ts[5:]) * 1e6)
    print(f"| | {len(devs) if devs else 1} | {tb:.2f} | {res[0]:.0f} | {res[1]:.0f} | {res[2]:.0f} |", flush=True)
