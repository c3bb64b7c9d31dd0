"""Looks for performance cliffs off the beaten path: a grid of store sizes x batch sizes x k x result modes x filters / masks,
AUTO path, wall per call (best of 3 after a warm-up) against a floor of (passes over the store at 6.5 TB/s + 40 us); prints
every cell and marks the ones more than 3x above their floor.  Not a benchmark of record: a smoke detector."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Cmp, Metric, VecStore

dim = 768
rng = np.random.default_rng(0)
print("| rows | nq | k | mode | variant | ms | floor ms | x floor | path | passes |")
print("|---|---|---|---|---|---|---|---|---|---|")
worst = []
for n in (10_000, 1_000_000, 10_000_000):
    s = VecStore(dim); s.set_chunk_size(4096); s.reserve(n); s.append_random(n, 5)
    row_mask = rng.random(n) < 0.5
    chunk_mask = rng.random((n + 4095) // 4096) < 0.5
    for nq in (1, 4, 16, 64):
        Q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
        for k in (10, 100, 1000, n):
            if k == n and (nq > 4 or n > 1_000_000):
                continue
            for perq in (False, True):
                for variant in ("plain", "filter", "row_mask", "chunk_mask", "l2", "take_min"):
                    if variant != "plain" and (k == n or (nq == 16)):
                        continue
                    metric = Metric.Euclidean if variant == "l2" else Metric.Cosine
                    def plan():
                        p = s.query(Q if nq > 1 else Q[0], metric)
                        p = p.take_min(k) if variant == "take_min" else p.take(k)
                        if perq: p = p.per_query()
                        if variant == "filter": p = p.filter(0.02, Cmp.Gt)
                        if variant == "row_mask": p = p.with_row_mask(row_mask)
                        return p
                    def run():
                        rq = plan().resolve()
                        return s._run(rq, chunk_mask=chunk_mask if variant == "chunk_mask" else None)
                    run()
                    best = 1e9
                    for _ in range(3):
                        t = time.perf_counter(); run(); best = min(best, time.perf_counter() - t)
                    st = s.last_stats if False else None
                    hits, counts, stats = run()
                    frac = 0.5 if variant == "chunk_mask" else 1.0
                    passes = max(int(stats["passes"]), 1) if stats else 1
                    floor = passes * frac * n * dim * 4 / 6.5e12 + 40e-6 + hits.size * 16 / 20e9  # + the hits' way to the host
                    ratio = best / floor
                    mark = " **<<**" if ratio > 3 else ""
                    print(f"| {n} | {nq} | {k} | {'perq' if perq else 'merged'} | {variant} | {best * 1e3:.3f} | {floor * 1e3:.3f} | {ratio:.1f}{mark} | {stats['path_used'] if stats else '-'} | {passes} |", flush=True)
                    worst.append((ratio, n, nq, k, perq, variant, best * 1e3))
    s.close()
worst.sort(reverse=True)
print("\nworst ten:")
for w in worst[:10]:
    print(w)
