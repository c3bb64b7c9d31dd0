"""Minimal driver for a kernel trace of the small-result path: rocprofv3 --kernel-trace --stats -- python3 benchmarks/default_take_trace.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from otters_amd import Metric, VecStore

s = VecStore(768)
s.append_random(10_000, 5)
q = np.random.default_rng(0).uniform(-1, 1, 768).astype(np.float32)
for _ in range(30):
    s.query(q, Metric.Cosine).collect_arrays()
s.close()
