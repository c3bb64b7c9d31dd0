#!/usr/bin/env python3
"""CPU baselines beside the GPU numbers of benchmarks/run_configs.py: the oracle (a C restatement of the reference's
loops, oracle/otters_oracle.c, built -O3 -mavx2) timed on this host on a bounded sample of each config.  VecStore
configs run single-threaded like src/vec.rs:223; the MetaStore config fans chunks out over all cores like rayon in
src/meta.rs:678.  Reported, never the product path."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

SEED = 0x07735
cores = os.cpu_count() or 1


def timed(fn, min_s=3.0, max_reps=50):
    fn()
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or (time.perf_counter() - t0 < min_s and reps < max_reps):
        fn()
        reps += 1
    return (time.perf_counter() - t0) / reps


rows_out = []


def report(name, sample, dt, units_per_call, full_over_sample, threads):
    rows_out.append((name, sample, threads, dt * 1e3, units_per_call / dt, units_per_call / dt / full_over_sample))


rng = np.random.default_rng(1)
# C1: the whole workload fits
r = O.rand_rows(0, 1_000_000, 128, SEED)
inv = O.inv_norms(r)
q = rng.uniform(-1, 1, (1, 128)).astype(np.float32)
dt = timed(lambda: O.vec_query(r, q, O.METRIC_DOT, O.TAKE_MAX, 10, inv=inv, fast=True))
report("C1 1M x 128 dot top-10, 1 query", "full", dt, 1, 1, 1)
del r

r = O.rand_rows(0, 1_000_000, 768, SEED)
inv = O.inv_norms(r)
q = rng.uniform(-1, 1, (1, 768)).astype(np.float32)
dt = timed(lambda: O.vec_query(r, q, O.METRIC_COSINE, O.TAKE_MAX, 10, inv=inv, fast=True))
report("headline 10M x 768 cosine top-10, 1 query", "1M rows (1/10)", dt, 1, 10, 1)

Q = rng.uniform(-1, 1, (256, 768)).astype(np.float32)
dt = timed(lambda: O.vec_query(r[:100_000], Q, O.METRIC_COSINE, O.TAKE_MAX, 100, inv=inv[:100_000], fast=True), min_s=5.0)
report("C2 10M x 768 cosine top-100, 256 queries (merged)", "100k rows (1/100)", dt, 256, 100, 1)

cs = 4096
n_chunks = (r.shape[0] + cs - 1) // cs
cm = (np.arange(n_chunks) % 2) == 1
dt = timed(lambda: O.meta_query(r, cs, q, O.METRIC_COSINE, O.TAKE_MAX, 10, filter_cmp=O.CMP_GT, filter_thr=0.5, chunk_mask=cm,
                                n_threads=cores, inv=inv, fast=True))
report("C3 10M x 768 MetaStore chunk 4096, half the chunks pruned, vec_filter(0.5,Gt), top-10", "1M rows (1/10)", dt, 1, 10, cores)

print("| config | sample | threads | ms per call on the sample | queries/s on the sample | queries/s extrapolated to the full config |")
print("|---|---|---|---|---|---|")
for name, sample, th, ms, qps, qps_full in rows_out:
    print(f"| {name} | {sample} | {th} | {ms:.1f} | {qps:.2f} | {qps_full:.3f} |")
