import sys; sys.path.insert(0,'.')
import numpy as np, oracle as O
from otters_amd import *
rng=np.random.default_rng(0)
for dim in (3, 8, 96):
    rows=rng.uniform(-1,1,(300,dim)).astype(np.float32); q=rng.uniform(-1,1,(1,dim)).astype(np.float32)
    s=VecStore(dim); s.add_vectors(rows)
    gi=s.inv_norms(); oi=O.inv_norms(rows)
    print(dim,"inv mismatch", (gi.view(np.uint32)!=oi.view(np.uint32)).sum(), gi[:3], oi[:3])
    for m,name in ((2,'dot'),(1,'l2'),(0,'cos')):
        plan=s.query(q, Metric(m)).take_max(300)
        rq=plan.resolve(); hits,_,_=s._run(rq)
        ref=O.vec_query(rows,q,m,1,300,ties=1)
        g=dict(zip(hits['index'].tolist(), hits['score'].view(np.uint32).tolist())); r=dict(zip(ref['index'].tolist(), ref['score'].view(np.uint32).tolist()))
        bad=[i for i in r if g.get(i)!=r[i]]
        print(dim,name,"len",len(hits),len(ref),"score-bit mismatches",len(bad), [(i, hex(g.get(i,0)), hex(r[i])) for i in bad[:3]])
