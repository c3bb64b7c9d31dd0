"""The batch cascade on clustered corpora at 10M x 768 (ott_store_append_clustered): per regime, 20 consecutive batches on a
fresh store — wall and score-phase time, how far down the cascade the batch went (refined = queries the hi pass could not
certify, retries = queries answered by the exact path, gate_failed = speculative gates that turned out too tight), corpus
passes — which also shows the store's back-off (hi pass skipped after it kept failing).  Results are exact at every level."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402  (query generation only: unseen members of the clusters)
from otters_amd import Metric, VecStore  # noqa: E402

N, DIM, SEED = int(os.environ.get("OTT_N", 10_000_000)), 768, 0xC1A57E
REGIMES = [("uniform (reference point)", None), ("4096 clusters, spread 0.45, aniso 2", (4096, 0.45, 2.0)), ("1000 clusters, spread 0.25", (1000, 0.25, 0.0)),
           ("20000 clusters, spread 0.10", (20_000, 0.10, 0.0)), ("100000 clusters, spread 0.04 (near-duplicates)", (100_000, 0.04, 0.0))]
for name, reg in REGIMES:
    for nq, k in ((64, 10), (256, 100)):
        store = VecStore(DIM)
        store.reserve(N)
        if reg is None:
            store.append_random(N, SEED)
            Q = O.rand_rows(0, nq * 20, DIM, SEED + 1)
        else:
            store.append_clustered(N, SEED, *reg)
            Q = O.clustered_rows(N + 7, nq * 20, DIM, SEED, *reg)
        store.prepare_batch()
        print(f"\n### {name}: {nq} queries, top-{k}\n")
        print("| batch | wall ms | score ms | passes | i8_refined | refined | retries | gate_failed | rescored |")
        print("|---|---|---|---|---|---|---|---|---|")
        walls = []
        for b in range(20):
            q = Q[b * nq:(b + 1) * nq]
            t = time.perf_counter()
            store.query(q, Metric.Cosine).take(k).collect_arrays()
            dt = (time.perf_counter() - t) * 1e3
            st = store.last_stats
            walls.append(dt)
            if b < 6 or b % 4 == 3:
                print(f"| {b} | {dt:.2f} | {st['score_ns'] / 1e6:.2f} | {st['passes']} | {st['i8_refined']} | {st['refined']} | {st['retries']} | {st['gate_failed']} | {st['rescored']} |", flush=True)
        print(f"\nmedian of batches 4..19: {np.median(walls[4:]):.2f} ms; exact path for the same batch would take ~{(nq + 3) // 4 * 4.5:.0f} ms", flush=True)
        store.close()
