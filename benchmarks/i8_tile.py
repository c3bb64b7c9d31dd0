"""One library build (OTT_LIB_PATH) on config 2's first cascade level: 10M x 768, 256 queries, cosine top-100 through the batch path.
Prints the score phase (median of 9), a checksum of the result, and — when the build is the diagnostic one — the in-kernel stamps of the
candidate kernel's tile (prologue / K loop / epilogue cycles, waits).  Used by benchmarks/i8_tile_variants.sh to compare experiment
builds of ott_mfma.hip (otters_amd/csrc/variants/build.sh).

    OTT_LIB_PATH=otters_amd/csrc/variants/lib_<name>.so python benchmarks/i8_tile.py [rows] [queries] [hi_fmt]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, Path, VecStore

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
fmt = int(sys.argv[3]) if len(sys.argv) > 3 else -1
s = VecStore(768)
s.set_option("hi_fmt", fmt)
s.reserve(rows)
s.append_random(rows, 5)
q = np.random.default_rng(1).uniform(-1, 1, (nq, 768)).astype(np.float32)
run = lambda: s.query(q, Metric.Cosine).take(100).with_path(Path.Mfma).collect_arrays()
for _ in range(3):
    hits, _ = run()
sc, wl = [], []
for _ in range(9):
    t0 = time.perf_counter()
    hits, _ = run()
    wl.append((time.perf_counter() - t0) * 1e3)
    sc.append(s.last_stats["score_ns"] / 1e6)
st = s.last_stats
h = hashlib.sha256(hits["index"].tobytes() + hits["score"].tobytes() + hits["query"].tobytes()).hexdigest()[:16]
print(f"RESULT lib={os.path.basename(os.environ.get('OTT_LIB_PATH', 'default'))} rows={rows} nq={nq} fmt={fmt} score_ms={np.median(sc):.3f} min={np.min(sc):.3f} "
      f"wall_ms={np.median(wl):.3f} i8_refined={st['i8_refined']} refined={st['refined']} retries={st['retries']} violations={st['bound_violations']} sha={h}", flush=True)
try:
    s.set_option("mfma_debug", 1)
except Exception:
    sys.exit(0)
# timing ablations (diagnostic build, option mfma_abl; results are garbage, the stamps and the score phase are what is read):
# 16 the query pieces of every third tile left out, 32 no query pieces
for abl in [int(a) for a in os.environ.get("ABL", "0").split(",")]:
    s.set_option("mfma_abl", abl)
    print(f"ABL {abl}", file=sys.stderr, flush=True)
    sc = []
    for _ in range(4):
        run()
        sc.append(s.last_stats["score_ns"] / 1e6)
    print(f"ABL {abl} stamped-kernel score_ms={min(sc):.3f}", file=sys.stderr, flush=True)
