"""Round 6: the int8 level re-scoring 4096 candidates per query (the wide finalize) on the clustered regimes of clustered_10m.py,
against the default ladder (int8 at 512 -> back-off -> half hi pass).  The experiment's library (force_fallback bit 128, `wide` allowed
at the int8 level, a 4096 rung in ott_api.hip's ladder) did NOT stay in the tree — it measured slower than the half pass
(profiles/round6/clustered.md holds the diff's description and the numbers); with the current library only the default mode runs.  Per regime and batch shape: median wall / score-phase ms over
batches 4..15, how many queries each level left open, and that both modes return the same bits."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
from otters_amd import Metric, VecStore  # noqa: E402

N, DIM, SEED = int(os.environ.get("OTT_N", 10_000_000)), 768, 0xC1A57E
REGIMES = [("uniform", None), ("4096 clusters, spread 0.45, aniso 2", (4096, 0.45, 2.0)), ("1000 clusters, spread 0.25", (1000, 0.25, 0.0)),
           ("20000 clusters, spread 0.10", (20_000, 0.10, 0.0)), ("100000 clusters, spread 0.04", (100_000, 0.04, 0.0))]
MODES = [("default ladder", None), ("int8 at 4096 from the first call", 128)]
if os.environ.get("OTT_MODES") == "default":
    MODES = MODES[:1]
print("| regime | batch | mode | wall ms | score ms | passes | i8 open | hi open | exact | rescored | same bits |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name, reg in REGIMES:
    store = VecStore(DIM)
    store.reserve(N)
    if reg is None:
        store.append_random(N, SEED)
    else:
        store.append_clustered(N, SEED, *reg)
    store.prepare_batch()
    for nq, k in ((256, 100), (64, 10)):
        Q = O.rand_rows(0, nq * 16, DIM, SEED + 1) if reg is None else O.clustered_rows(N + 7, nq * 16, DIM, SEED, *reg)
        first = {}
        for mode, bit in MODES:
            try:
                store.set_option("force_fallback", bit or 0)
            except Exception:  # noqa: BLE001 -- the experiment's bit is not in this library
                continue
            walls, scores, last = [], [], None
            for b in range(16):
                q = Q[b * nq:(b + 1) * nq]
                t = time.perf_counter()
                hits, _ = store.query(q, Metric.Cosine).take(k).per_query().collect_arrays()
                walls.append((time.perf_counter() - t) * 1e3)
                st = store.last_stats
                scores.append(st["score_ns"] / 1e6)
                if b == 0:
                    first[mode] = hits.copy()
                last = st
            same = "" if len(first) < 2 else str(bool(np.array_equal(first[MODES[0][0]]["index"], first[mode]["index"]) and
                                                      np.array_equal(first[MODES[0][0]]["score"].view(np.uint32), first[mode]["score"].view(np.uint32))))
            print(f"| {name} | {nq} x top-{k} | {mode} | {np.median(walls[4:]):.2f} | {np.median(scores[4:]):.2f} | {last['passes']} | {last['i8_refined']} | "
                  f"{last['refined']} | {last['retries']} | {last['rescored']} | {same} |", flush=True)
    store.close()
