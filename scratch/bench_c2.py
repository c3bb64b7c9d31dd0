import sys, time; sys.path.insert(0,'.')
import numpy as np
from otters_amd import *
n=int(sys.argv[1]) if len(sys.argv)>1 else 10_000_000
nq=int(sys.argv[2]) if len(sys.argv)>2 else 256
k=int(sys.argv[3]) if len(sys.argv)>3 else 100
dim=768
s=VecStore(dim); s.reserve(n); s.append_random(n, 0x7735)
q=np.random.default_rng(1).uniform(-1,1,(nq,dim)).astype(np.float32)
for it in range(3):
    t=time.perf_counter()
    res=s.query(q, Metric.Cosine).take(k).with_path(Path.Mfma).per_query().collect()
    dt=time.perf_counter()-t
    st=s.last_stats
    fl=2.0*n*nq*dim
    print(f"iter{it}: wall {dt*1e3:.1f} ms  score {st['score_ns']/1e6:.2f} ms  final {st['merge_ns']/1e6:.2f} ms  retries {st['retries']}  TF/s(score) {fl/st['score_ns']/1e3:.1f}  frac {fl/st['score_ns']/1e3/157.3:.3f}")
# cross-check 8 queries against the exact path
ex=s.query(q[:8], Metric.Cosine).take(k).with_path(Path.Exact).per_query().collect()
ok=all([ (a.index,a.score) for a in ex[i]]==[(b.index,b.score) for b in res[i]] for i in range(8))
print("exact-path cross-check on 8 queries:", ok, s.last_stats['score_ns']/1e6)
m=s.query(q, Metric.Cosine).take(k).with_path(Path.Mfma).collect()
print("merged top:", m[:3], len(m))
