"""How dense are the clustered regimes around the k-th score?  For each regime of benchmarks/clustered_10m.py: 16 queries, the exact
top-16384 per query (exact-order path), and how many rows lie within d of the 100th best score for d = the int8 level's bound
(8e-3), a quarter and a sixteenth of it, and the half plane's (2.1e-4).  This is what decides how many candidates a level must
re-score to certify: rows within its bound of the k-th score, plus k."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
from otters_amd import Metric, Path, VecStore  # noqa: E402

N, DIM, SEED = int(os.environ.get("OTT_N", 10_000_000)), 768, 0xC1A57E
REGIMES = [("uniform", None), ("4096 clusters, spread 0.45, aniso 2", (4096, 0.45, 2.0)), ("1000 clusters, spread 0.25", (1000, 0.25, 0.0)),
           ("20000 clusters, spread 0.10", (20_000, 0.10, 0.0)), ("100000 clusters, spread 0.04", (100_000, 0.04, 0.0))]
K, DEEP, NQ = 100, 16384, 16
print("| regime | 100th score (median) | rows within 8e-3 | within 2e-3 | within 5e-4 | within 2.1e-4 | within 6e-5 | (median / max over 16 queries; 16384 = at least) |")
print("|---|---|---|---|---|---|---|---|")
for name, reg in REGIMES:
    store = VecStore(DIM)
    store.reserve(N)
    if reg is None:
        store.append_random(N, SEED)
        Q = O.rand_rows(0, NQ, DIM, SEED + 1)
    else:
        store.append_clustered(N, SEED, *reg)
        Q = O.clustered_rows(N + 7, NQ, DIM, SEED, *reg)
    hits, counts = store.query(Q, Metric.Cosine).take(DEEP).per_query().with_path(Path.Exact).collect_arrays()
    sc = hits["score"].reshape(NQ, DEEP)
    kth = sc[:, K - 1]
    cols = []
    for d in (8e-3, 2e-3, 5e-4, 2.1e-4, 6e-5):
        cnt = (sc >= (kth - d)[:, None]).sum(1) - K
        cols.append(f"{int(np.median(cnt))} / {int(cnt.max())}")
    print(f"| {name} | {np.median(kth):.4f} | " + " | ".join(cols) + " | |", flush=True)
    store.close()
