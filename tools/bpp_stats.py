import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, smallk_amd, oracle
smallk_amd.initialize(0)
for (m, n, k) in ((16384, 4096, 64), (8192, 4096, 16)):
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43); H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    for it in (1, 3, 8):
        r = smallk_amd.nmf(A, W0, H0, "BPP", min_iter=it, max_iter=it, normalize=False)
        print(m, n, k, "iters", it, "zero frac W %.3f H %.3f" % ((r.W == 0).mean(), (r.H == 0).mean()),
              "cols of H with no zero %.3f" % ((r.H > 0).all(axis=0).mean()), "rows of W with no zero %.3f" % ((r.W > 0).all(axis=1).mean()))
