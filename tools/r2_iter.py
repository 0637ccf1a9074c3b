"""RANK2 iterations on a C5-shaped sparse matrix (for rocprofv3 kernel tables): python tools/r2_iter.py [nodes] [deg] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import smallk_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rng = np.random.default_rng(0)
nh = n * deg // 2
src = rng.integers(0, n, size=nh); dst = rng.integers(0, n, size=nh)
A = sp.coo_matrix((np.ones(nh), (src, dst)), shape=(n, n)); A = (A + A.T).tocsc(); A.sum_duplicates()
smallk_amd.initialize(0)
W0 = smallk_amd.uniform_host(n, 2, 43); H0 = smallk_amd.uniform_host(2, n, 44)
for rep in range(2):
    t0 = time.time()
    r = smallk_amd.nmf_sparse(A, W0, H0, "RANK2", min_iter=iters, max_iter=iters, tol=1e-4)
    print(f"rep {rep}: {r.iteration_count} iterations, {r.elapsed_us/ max(r.iteration_count,1):.1f} us each (solver clock), wall {time.time()-t0:.2f}s", flush=True)
