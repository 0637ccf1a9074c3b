"""where the wall time of the C5-shaped HierNMF2 run goes: matrix creation (host transpose + upload) vs the clustering"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import smallk_amd
from smallk_amd import solver as S
n, deg = 1_000_000, 16
rng = np.random.default_rng(0)
comm = rng.integers(0, 16, size=n)
order = np.argsort(comm, kind="stable")
starts = np.searchsorted(comm[order], np.arange(17))
nnz_half = n * deg // 2
src = rng.integers(0, n, size=nnz_half); intra = rng.random(nnz_half) < 0.85; dst = rng.integers(0, n, size=nnz_half)
c = comm[src[intra]]
dst[intra] = order[starts[c] + (rng.random(intra.sum()) * (starts[c + 1] - starts[c])).astype(np.int64)]
A = sp.coo_matrix((np.ones(nnz_half), (src, dst)), shape=(n, n)); A = (A + A.T).tocsc(); A.sum_duplicates()
smallk_amd.initialize(0)
for rep in range(2):
    t0 = time.time(); M = S.SparseMatrix.from_scipy(A); t1 = time.time()
    print(f"rep {rep}: SparseMatrix.from_scipy (host transpose + upload): {t1-t0:.2f} s", flush=True)
    M.close()
    t0 = time.time(); res = smallk_amd.hier_nmf2(A, 8, seed=1, tol=1e-4, max_iter=5000, verbose=False); t1 = time.time()
    print(f"rep {rep}: hier_nmf2 end to end {t1-t0:.2f} s", flush=True)
