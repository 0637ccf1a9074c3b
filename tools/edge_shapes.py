"""edge shapes vs the oracle (tiny / degenerate dimensions)"""
import sys
sys.path.insert(0, '.')
import numpy as np
import oracle, smallk_amd
smallk_amd.initialize(0)
def rel(a,b): return np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300)
bad = 0
for (m,n,k) in [(1,1,1),(1,5,1),(5,1,1),(3,2,2),(2,3,2),(7,7,7),(64,64,64),(65,63,33),(129,1,1),(1,300,1),(257,130,17),(1000,9,9),(9,1000,9),(130,258,64)]:
    for alg in ("MU","HALS","BPP","RANK2"):
        if alg == "RANK2" and k != 2: continue
        A = oracle.fill_uniform(m,n,42) + 0.01
        W0 = oracle.fill_uniform(m,k,43) + 0.01; H0 = oracle.fill_uniform(k,n,44) + 0.01
        for st in ("f32","bf16"):
            Aq = oracle.quantize(A, 1 if st=="bf16" else 0)
            r = oracle.nmf(Aq,W0,H0,alg,min_iter=4,max_iter=4)
            g = smallk_amd.nmf(Aq,W0,H0,alg,min_iter=4,max_iter=4,storage=st)
            if r.result != g.result:
                print("RESULT MISMATCH",m,n,k,alg,st,r.result,g.result); bad += 1; continue
            if r.result == 0:
                e = max(rel(g.W,r.W),rel(g.H,r.H))
                if not (e < 1e-4) : print("PARITY",m,n,k,alg,st,e); bad += 1
print("edge shapes done, failures:", bad)
