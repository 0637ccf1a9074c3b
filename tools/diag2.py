import time, sys, os, subprocess
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np
import oracle, smallk_amd, make_golden as mg
smallk_amd.initialize(0)
def rel(a,b): return np.linalg.norm(a-b)/np.linalg.norm(b)
for (m,n,k,pl) in [(300,200,33,True),(256,192,64,True),(512,256,8,False),(2048,1024,32,False)]:
    A = mg.make_A(m,n,k,pl,1) if pl else oracle.fill_uniform(m,n,42,quant=1)
    W0=oracle.fill_uniform(m,k,43); H0=oracle.fill_uniform(k,n,44)
    for alg in ("MU","HALS","BPP"):
        for it in (5,20):
            r=oracle.nmf(A,W0,H0,alg,min_iter=it,max_iter=it)
            out=[]
            for st in ("f32","bf16"):
                g=smallk_amd.nmf(A,W0,H0,alg,min_iter=it,max_iter=it,storage=st)
                out.append("%s W %.1e H %.1e"%(st, rel(g.W,r.W), rel(g.H,r.H)))
            print(m,n,k,alg,it," | ".join(out), flush=True)
