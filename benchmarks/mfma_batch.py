import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from otters_amd import *
nq=int(sys.argv[1]); n=int(sys.argv[2]) if len(sys.argv)>2 else 10_000_000
s=VecStore(768); s.reserve(n); s.append_random(n, 5)
q=np.random.default_rng(1).uniform(-1,1,(nq,768)).astype(np.float32)
for it in range(3):
    t=time.perf_counter(); r=s.query(q,Metric.Cosine).take(100).with_path(Path.Mfma).collect(); dt=time.perf_counter()-t
st=s.last_stats; print("nq",nq,"wall %.3f ms"%(dt*1e3),"score %.3f ms"%(st["score_ns"]/1e6))
