import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Metric, Path, VecStore
dim, n = 768, 10_000_000
s = VecStore(dim); s.set_option("hi_prebuild", 0); s.reserve(n); s.append_random(n, 5)
qs = np.random.default_rng(1).uniform(-1, 1, (64, dim)).astype(np.float32)
print("| threads | queries/s | mean latency ms |")
for nt in (1, 2, 4, 8):
    per = 40
    def work(i):
        for j in range(per):
            s.query(qs[(i * 7 + j) % 64], Metric.Cosine).take(10).with_path(Path.Exact).collect_arrays()
    work(0)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    t = time.perf_counter()
    [th.start() for th in ths]; [th.join() for th in ths]
    dt = time.perf_counter() - t
    print(f"| {nt} | {nt * per / dt:.1f} | {dt / per * 1e3:.2f} |", flush=True)
