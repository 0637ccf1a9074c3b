import sys
sys.path.insert(0, '.')
import numpy as np
import oracle, smallk_amd
smallk_amd.initialize(0)
m=n=k=7
A = oracle.fill_uniform(m,n,42) + 0.01
W0 = oracle.fill_uniform(m,k,43) + 0.01; H0 = oracle.fill_uniform(k,n,44) + 0.01
for it in (1,2,3,4):
    r = oracle.nmf(A,W0,H0,"BPP",min_iter=it,max_iter=it,normalize=False)
    g = smallk_amd.nmf(A,W0,H0,"BPP",min_iter=it,max_iter=it,storage="f32",normalize=False)
    print(it, r.result, g.result, g.iteration_count, np.linalg.cond(r.W.T@r.W), np.linalg.cond(r.H@r.H.T))
    if g.result==0: print("   err", np.linalg.norm(g.W-r.W)/np.linalg.norm(r.W))
print("cond W0tW0", np.linalg.cond(W0.T@W0))
