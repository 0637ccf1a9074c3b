import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
import smallk_amd, oracle
smallk_amd.initialize(0)
for n in (300000, 500000, 1000000):
    rng = np.random.default_rng(0)
    nnz_half = n * 8
    src = rng.integers(0, n, size=nnz_half); dst = rng.integers(0, n, size=nnz_half)
    A = sp.coo_matrix((np.ones(nnz_half), (src, dst)), shape=(n, n)); A = (A + A.T).tocsc(); A.sum_duplicates()
    W0 = oracle.fill_uniform(n, 2, 1); H0 = oracle.fill_uniform(2, n, 2)
    for it in (1, 2, 5):
        r = smallk_amd.nmf_sparse(A, W0, H0, "RANK2", min_iter=it, max_iter=it, tol=1e-4)
        print(n, it, "rc", r.result, "iters", r.iteration_count, "W finite", np.isfinite(r.W).all(), "H finite", np.isfinite(r.H).all(),
              "Wmax", np.nanmax(np.abs(r.W)), "Hmax", np.nanmax(np.abs(r.H)), flush=True)
