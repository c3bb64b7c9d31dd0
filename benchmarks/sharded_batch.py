"""One rank's share of BASELINE config 4 through the SHARDED code path (ott_query_sharded: score -> ncclAllGather -> merge on one
stream, behind the C ABI) as a 1-rank RCCL communicator on one GPU, beside the same batch through the plain store: what the
exchange path costs on top of scoring.  (The 8-GPU run itself is the driver's; this only shows the per-rank overhead.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otters_amd import Metric, VecStore  # noqa: E402
from otters_amd.dist import Comm, ShardedVecStore  # noqa: E402

comm = Comm.rccl(Comm.unique_id(), 0, 1, 0)
n, dim = int(os.environ.get("ROWS", "5000000")), 768
store = VecStore(dim)
store.reserve(n)
store.append_random(n, 0x07735)
sh = ShardedVecStore(store, comm)
rng = np.random.default_rng(4)
print("| queries | mode | plain store ms | sharded path ms |")
print("|---|---|---|---|")
for nq in (1, 256, 1024):
    Q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    for perq in (False, True):
        def plan(o):
            p = o.query(Q if nq > 1 else Q[0], Metric.Cosine).take(100 if nq > 1 else 10)
            return p.per_query() if perq else p
        res = []
        for o in (store, sh):
            plan(o).collect_arrays()
            t = []
            for _ in range(5):
                t0 = time.perf_counter()
                plan(o).collect_arrays()
                t.append(time.perf_counter() - t0)
            res.append(np.median(t) * 1e3)
        print(f"| {nq} | {'per-query' if perq else 'merged'} | {res[0]:.3f} | {res[1]:.3f} |", flush=True)
comm.close()
