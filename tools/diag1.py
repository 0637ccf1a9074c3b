import time, sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np
import oracle, smallk_amd, make_golden as mg
print("cpus", os.cpu_count(), "omp threads", oracle.num_threads())
smallk_amd.initialize(0)
m,n,k = 300,200,33
A = mg.make_A(m,n,k,True,0); W0=oracle.fill_uniform(m,k,43); H0=oracle.fill_uniform(k,n,44)
for alg in ("MU","HALS","BPP"):
    for thr in (0, 8):
        t=time.time(); r=oracle.nmf(A,W0,H0,alg,min_iter=20,max_iter=20,max_threads=thr); t1=time.time()-t
        print(alg, "oracle threads", thr, "%.3fs"%t1)
    t=time.time(); g=smallk_amd.nmf(A,W0,H0,alg,min_iter=20,max_iter=20); t2=time.time()-t
    print(alg, "gpu %.3fs"%t2, np.linalg.norm(g.W-r.W)/np.linalg.norm(r.W))
    t=time.time(); g=smallk_amd.nmf(A,W0,H0,alg,min_iter=20,max_iter=20); t2=time.time()-t
    print(alg, "gpu again %.3fs"%t2)
