"""quick parity screen for kernel variants: a few shapes x algorithms vs the oracle"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np
import oracle, smallk_amd, make_golden as mg
smallk_amd.initialize(0)
def rel(a,b): return np.linalg.norm(a-b)/np.linalg.norm(b)
worst=0
for (m,n,k,pl,q,st) in [(300,200,33,True,1,"bf16"),(256,192,64,True,0,"f32"),(2048,1024,32,False,1,"bf16"),(1000,3000,20,False,1,"bf16"),(4096,512,16,False,0,"f32")]:
    A = mg.make_A(m,n,k,pl,q) if pl else oracle.fill_uniform(m,n,42,quant=q)
    W0=oracle.fill_uniform(m,k,43); H0=oracle.fill_uniform(k,n,44)
    for alg in ("HALS","BPP"):
        r=oracle.nmf(A,W0,H0,alg,min_iter=5,max_iter=5)
        g=smallk_amd.nmf(A,W0,H0,alg,min_iter=5,max_iter=5,storage=st)
        e=max(rel(g.W,r.W),rel(g.H,r.H)); worst=max(worst,e)
print("variant", os.environ.get("SMK_BP_VARIANT","-"), "worst rel err %.2e"%worst, "OK" if worst<1e-4 else "FAIL")
