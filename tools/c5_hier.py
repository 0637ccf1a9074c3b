"""C5-shaped run: HierNMF2 on a synthetic sparse symmetric adjacency (planted communities).
usage: python tools/c5_hier.py [nodes] [avg_degree] [clusters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import smallk_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 16
clusters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rng = np.random.default_rng(0)
t0 = time.time()
comm = rng.integers(0, 16, size=n)
order = np.argsort(comm, kind="stable")
starts = np.searchsorted(comm[order], np.arange(17))
nnz_half = n * deg // 2
src = rng.integers(0, n, size=nnz_half)
intra = rng.random(nnz_half) < 0.85
dst = rng.integers(0, n, size=nnz_half)
c = comm[src[intra]]
dst[intra] = order[starts[c] + (rng.random(intra.sum()) * (starts[c + 1] - starts[c])).astype(np.int64)]
A = sp.coo_matrix((np.ones(nnz_half), (src, dst)), shape=(n, n))
A = (A + A.T).tocsc()
A.sum_duplicates()
print(f"graph: {n} nodes, nnz {A.nnz}, build {time.time()-t0:.1f}s", flush=True)
smallk_amd.initialize(0)
t0 = time.time()
res = smallk_amd.hier_nmf2(A, clusters, seed=1, tol=1e-4, max_iter=5000, verbose=True)
dt = time.time() - t0
asg = res.get_assignments()
print(f"hier_nmf2: {dt:.2f}s  factorizations {res.nmf_count} (hit max_iter: {res.max_count}), outliers {len(res.get_outliers())}")
ok = asg != 0xFFFFFFFF
pur = 0
for leaf in np.unique(asg[ok]):
    pur += np.bincount(comm[asg == leaf], minlength=16).max()
print(f"leaves {len(np.unique(asg[ok]))}, purity {pur/ok.sum():.3f}")
