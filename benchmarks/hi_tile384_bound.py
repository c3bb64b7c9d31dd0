"""Round 4, verdict item 4: what could a 384 x 256 tile gain on config 2's hi pass?  Measured, not argued: in the DIAGNOSTIC
build (make OUT=libotters_hip_dbg.so OBJDIR=_obj_dbg EXTRA=-DOTT_MFMA_DEBUG_BUILD; OTT_LIB_PATH points at it) option mfma_abl = 16
leaves the query pieces of every THIRD row tile out: the L2 -> LDS query traffic per corpus row is then that of a 384-row tile
(2/3 of today's), everything else — HBM row stream, matrix loop, epilogue — is today's (and the counted waits return early
for those tiles, which flatters the ablated number further).  mfma_abl = 32 leaves ALL query pieces out: the floor of the
fill path.  No real 384 x 256 kernel can beat the first number (it would also run ONE wave per SIMD: 384 accumulators per
lane).  The ablated scores are garbage, the certification notices (bound_violations) and answers on the exact path — so the
figure to read is the candidate pass's own kernel time: run each mode under rocprofv3 --kernel-trace and sum the
mfma_score_kernel dispatches of a batch (benchmarks/hi_tile384_bound.sh does).

    OTT_LIB_PATH=otters_amd/csrc/libotters_hip_dbg.so python benchmarks/hi_tile384_bound.py <abl> [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, Path, VecStore

abl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
s = VecStore(768)
s.set_option("hi_prebuild", 0)
s.reserve(n)
s.append_random(n, 5)
q = np.random.default_rng(1).uniform(-1, 1, (nq, 768)).astype(np.float32)
s.set_option("mfma_debug", 1)  # the stamped kernel variants (every mode pays for the stamps alike)
s.set_option("mfma_abl", abl)
for it in range(4):
    s.query(q, Metric.Cosine).take(100).with_path(Path.Mfma).collect_arrays()
print("abl", abl, "bound_violations", s.last_stats["bound_violations"], "retries", s.last_stats["retries"])
