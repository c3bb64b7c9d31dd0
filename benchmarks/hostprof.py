"""Host-side phases of a large batch (diagnostic build only: OTT_LIB_PATH=.../libotters_hip_dbg.so, option mfma_debug): the library
prints prepare / enqueue / wait / unpack milliseconds per candidate pass on stderr."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from otters_amd import Metric, VecStore
for n, nq, k in ((10_000_000, 256, 100), (5_000_000, 1024, 100)):
    s = VecStore(768); s.set_option("mfma_debug", 1); s.reserve(n); s.append_random(n, 5)
    q = np.random.default_rng(1).uniform(-1, 1, (nq, 768)).astype(np.float32)
    for it in range(4):
        t = time.perf_counter(); s.query(q, Metric.Cosine).take(k).collect_arrays(); dt = time.perf_counter() - t
        print(n, nq, "wall %.3f ms" % (dt * 1e3), "score %.3f" % (s.last_stats["score_ns"] / 1e6), "merge %.3f" % (s.last_stats["merge_ns"] / 1e6), flush=True)
    s.close()
