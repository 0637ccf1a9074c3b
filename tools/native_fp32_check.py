"""SMK_NSPLIT=1: fp32 A on the native fp32 matrix cores (v_mfma_f32_32x32x2_f32, no emulation): one iteration against the
oracle.  (The fp32 accumulation of this form is 10x less accurate than the emulated ones; ill-conditioned cases of
tools/quick_parity.py amplify that beyond the parity bar after a few iterations, which is why it is not the default.)"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np
import oracle, smallk_amd, make_golden as mg
smallk_amd.initialize(0)
def rel(a,b): return np.linalg.norm(a-b)/np.linalg.norm(b)
for (m,n,k,pl) in [(256,192,64,True),(4096,512,16,False),(300,200,33,True),(512,256,8,False),(96,64,5,False)]:
    A = mg.make_A(m,n,k,pl,0) if pl else oracle.fill_uniform(m,n,42,quant=0)
    W0=oracle.fill_uniform(m,k,43); H0=oracle.fill_uniform(k,n,44)
    for alg in ("MU","HALS"):
        r=oracle.nmf(A,W0,H0,alg,min_iter=1,max_iter=1)
        g=smallk_amd.nmf(A,W0,H0,alg,min_iter=1,max_iter=1,storage="f32")
        print(m,n,k,alg,"W %.2e H %.2e"%(rel(g.W,r.W),rel(g.H,r.H)))
        assert rel(g.W,r.W) < 1e-5 and rel(g.H,r.H) < 1e-5
print("OK")
