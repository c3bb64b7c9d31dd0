#!/usr/bin/env python3
"""Times the five BASELINE.json configs that fit one GPU (C0-C3 + the headline single-query
cosine) and prints a markdown table + one JSON object.  Not the driver's bench (that is
bench.py); this is the evidence behind DESIGN.md / profiles/."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from otters_amd import Cmp, Column, DataType, MetaStore, Metric, Mode, Path, VecStore, col  # noqa: E402

HBM, MFMA_BF16, MFMA_F32, MFMA_I8 = 8000.0, 2500.0, 157.3, 5000.0  # HBM3E spec peak (GB/s), dense bf16 / f32 / int8 MFMA peaks (T(FL)OP/s): MI355X_MICROARCH.md
SEED = 0x07735
only = set(sys.argv[1:])


def timed(fn, reps):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        t.append(time.perf_counter() - t0)
    return out, float(np.median(t))


rows_out = []


def report(name, wall, stats, bytes_alg, flops=None, note=""):
    """bytes_alg: what the scoring launch must read — the f32 rows + inverse norms on the exact path; on the batch path the
    candidate pass reads the int8 plane instead (round 5: a quarter of the row bytes + one scale per row; every config here is
    cosine / dot with k <= 128, which the int8 level takes), and its operations run on the int8 matrix pipe."""
    k_ms = stats["score_ns"] / 1e6
    if stats["path_used"] == 2:
        bytes_alg = bytes_alg // 4 + (bytes_alg // (768 * 4 + 4)) * 4
    r = dict(config=name, wall_ms=round(wall * 1e3, 3), qps=None, score_kernel_ms=round(k_ms, 4), merge_ms=round(stats["merge_ns"] / 1e6, 4),
             hbm_GBs=round(bytes_alg / (k_ms * 1e-3) / 1e9, 1), hbm_frac=round(bytes_alg / (k_ms * 1e-3) / 1e9 / HBM, 4), note=note,
             path={1: "exact", 2: "mfma"}.get(stats["path_used"], "?"))
    if flops:
        r["TFLOPs"] = round(flops / (k_ms * 1e-3) / 1e12, 1)
        r["mfma_frac"] = round(flops / (k_ms * 1e-3) / 1e12 / MFMA_I8, 4)
        r["x_f32_peak"] = round(flops / (k_ms * 1e-3) / 1e12 / MFMA_F32, 2)  # SURVEY 8(d) prices the batch against the f32 matrix peak
    rows_out.append(r)
    print(r, flush=True)


if not only or "c1" in only:
    s = VecStore(128)
    s.append_random(1_000_000, SEED)
    q = np.random.default_rng(1).uniform(-1, 1, 128).astype(np.float32)
    res, w = timed(lambda: s.query(q, Metric.DotProduct).take(10).collect(), 50)
    report("C1 1Mx128 dot top-10, 1 query", w, s.last_stats, 1_000_000 * 128 * 4)
    rows_out[-1]["qps"] = round(1 / w, 1)
    s.close()

if not only or only & {"c2", "c3", "head"}:
    n, dim, cs = 10_000_000, 768, 4096
    bucket = Column.from_numpy("bucket", DataType.Int32, ((np.arange(n) // cs) % 2).astype(np.int32))
    meta = MetaStore.from_columns([bucket]).with_random_vectors(n, dim, SEED).with_chunk_size(cs).build()
    s = meta._store
    rng = np.random.default_rng(1)
    if not only or "head" in only:
        q = rng.uniform(-1, 1, dim).astype(np.float32)
        res, w = timed(lambda: s.query(q, Metric.Cosine).take(10).collect(), 30)
        report("HEAD 10Mx768 cosine top-10, 1 query", w, s.last_stats, n * (dim * 4 + 4))
        rows_out[-1]["qps"] = round(1 / w, 1)
    def config3(label):
        q = np.random.default_rng(3).uniform(-1, 1, dim).astype(np.float32)
        prng = np.random.default_rng(4)
        planted = np.arange(12_345, n, 156_007)[:64]
        for i in planted:
            s.write_rows(int(i), (q + prng.normal(0, 0.05, dim)).astype(np.float32)[None, :])
        def run():
            return meta.query(q, Metric.Cosine).meta_filter(col("bucket").eq(1)).vec_filter(0.5, Cmp.Gt).take(10).collect()
        res, w = timed(run, 30)
        st = meta.last_query_stats()
        scored = st.vectors_compared
        g = s.last_stats
        report(f"C3 10Mx768 MetaStore chunk 4096, 50% pruned, vec_filter(0.5,Gt), top-10{label}", w, g, scored * (dim * 4 + 4),
               note=f"pruned={st.pruned_chunks}/{st.total_chunks} hits={len(res)}")
        rows_out[-1]["qps"] = round(1 / w, 1)

    # config 3 first: no batch has run on this store, so AUTO sends the single query down the exact-order kernel
    if not only or "c3" in only:
        config3("")
    if not only or "c2" in only:
        Q = rng.uniform(-1, 1, (256, dim)).astype(np.float32)
        for mode, label in ((False, "merged"), (True, "per-query")):
            def run():
                p = s.query(Q, Metric.Cosine).take(100)
                return (p.per_query() if mode else p).collect_arrays()  # NumPy records (25 600 SearchResult objects cost ~8 ms)
            res, w = timed(run, 5)
            report(f"C2 10Mx768 cosine top-100, 256 queries ({label})", w, s.last_stats, n * (dim * 4 + 4), flops=2.0 * n * dim * 256,
                   note=f"refined={s.last_stats['refined']} retries={s.last_stats['retries']}")
            rows_out[-1]["qps"] = round(256 / w, 1)
    # ... and again once the bf16 hi plane is resident (config 2 built it): AUTO then answers a single query through the cascade
    if not only or ("c3" in only and "c2" in only):
        config3(" — hi plane resident")

if not only or "c4" in only:
    # C4 = 40M x 768 over 8 GPUs, 1024 queries, cosine top-100: one rank's share is 5M rows x 1024 queries (the
    # exchange afterwards is an all-gather of 1024 x 100 x 16 B = 1.6 MB per GPU + one grouped merge launch)
    n4, dim4 = 5_000_000, 768
    s4 = VecStore(dim4)
    s4.reserve(n4)
    s4.append_random(n4, SEED)
    Q4 = np.random.default_rng(4).uniform(-1, 1, (1024, dim4)).astype(np.float32)
    for mode, label in ((False, "merged"), (True, "per-query")):
        def run4():
            p = s4.query(Q4, Metric.Cosine).take(100)
            return (p.per_query() if mode else p).collect_arrays()
        res, w = timed(run4, 3)
        report(f"C4 shard (1 of 8 GPUs): 5Mx768 cosine top-100, 1024 queries ({label})", w, s4.last_stats, n4 * (dim4 * 4 + 4),
               flops=2.0 * n4 * dim4 * 1024, note=f"refined={s4.last_stats['refined']} retries={s4.last_stats['retries']} passes={s4.last_stats['passes']}")
        rows_out[-1]["qps"] = round(1024 / w, 1)
    s4.close()

print("\n| config | path | wall ms | score kernel ms | merge ms | GB/s (bytes the launch must read) | HBM frac | TOP/s (2 x dim x rows x queries) | frac of int8 MFMA peak (5000) | x the f32 MFMA peak (157.3) | q/s | note |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows_out:
    print(f"| {r['config']} | {r['path']} | {r['wall_ms']} | {r['score_kernel_ms']} | {r['merge_ms']} | {r['hbm_GBs']} | {r['hbm_frac']} | "
          f"{r.get('TFLOPs', '')} | {r.get('mfma_frac', '')} | {r.get('x_f32_peak', '')} | {r['qps']} | {r['note']} |")
print(json.dumps(rows_out))
