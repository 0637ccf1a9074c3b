"""Evidence for the block-pivoting kernels: how many DISTINCT passive sets a launch meets, how large the
sets are, how often a column's set changes between iterations, and -- from the device counters of csrc/nnls.hip
(SMK_NNLS_STATS=1, set below) -- exchanges per column and the size of every compact solve.

   python tools/nnls_sets.py <workload> [iterations]
   workloads: s_1m | s_reuters | c2 | c4s_uniform | c4s_planted | mid32_uniform | mid32_planted

The reference groups columns by identical passive set before factoring (BppSolveNormalEq, nmf_solver_bpp.hpp:29-142,
GroupIdenticalColumns bit_matrix.cpp:803-816); whether a grouped solve can pay on the device is decided by these counts."""
import os, sys
os.environ["SMK_NNLS_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import smallk_amd
from smallk_amd import _lib as L

name = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CHECK = sorted({1, 2, 3, 5, 10, 20, 30, 50} & set(range(1, iters + 1)) | {iters})
smallk_amd.initialize(0)


def masks(X):      # X: k x N (columns = the independent problems); returns one uint64 per column
    k = X.shape[0]
    w = (np.uint64(1) << np.arange(k, dtype=np.uint64))[:, None]
    return ((X > 0) * w).sum(axis=0, dtype=np.uint64)


def describe(tag, mk, prev, k):
    u, cnt = np.unique(mk, return_counts=True)
    cnt = np.sort(cnt)[::-1]
    N = mk.size
    pc = np.array([bin(int(x)).count("1") for x in u])
    # |F| histogram over columns
    sizes = np.zeros(k + 1, dtype=np.int64)
    _, inv = np.unique(mk, return_inverse=True)
    np.add.at(sizes, pc[inv], 1)
    nz = np.nonzero(sizes)[0]
    top = ", ".join(f"{c}" for c in cnt[:8])
    cover = [int(np.searchsorted(np.cumsum(cnt), f * N) + 1) for f in (0.5, 0.9, 0.99)]
    same = "" if prev is None else f"; unchanged since the previous checkpoint: {100.0 * float((prev == mk).mean()):.1f} %"
    print(f"   {tag}: {N} columns, {u.size} distinct sets ({100.0 * u.size / N:.2f} %); largest groups {top}; groups covering 50/90/99 % "
          f"of the columns: {cover[0]}/{cover[1]}/{cover[2]}{same}")
    print(f"      |F| over columns: min {nz[0]} max {nz[-1]} mean {float((sizes * np.arange(k + 1)).sum()) / N:.2f}; histogram "
          + " ".join(f"{i}:{sizes[i]}" for i in nz[:40]))


def stats(reset=True):
    out = (C.c_uint64 * 256)()
    rc = L.lib().smk_debug_nnls_stats(out, int(reset))
    assert rc == 0, rc
    return np.array(out[:], dtype=np.int64)


def show_stats(tag, st):
    cols = max(int(st[178]), 1)
    ex = st[0:16]
    first, later = st[16:81], st[96:161]
    print(f"   {tag}: {cols} column solves; exchanges per column " + " ".join(f"{i}:{ex[i]}" for i in range(16) if ex[i])
          + f" (mean {float((ex * np.arange(16)).sum()) / cols:.2f})")
    nzf, nzl = np.nonzero(first)[0], np.nonzero(later)[0]
    if nzf.size:
        print("      first solve size t: " + " ".join(f"{i}:{first[i]}" for i in nzf) + f" (mean {float((first * np.arange(65)).sum()) / max(first.sum(), 1):.2f})")
    if nzl.size:
        print("      later solve sizes t: " + " ".join(f"{i}:{later[i]}" for i in nzl) + f" (mean {float((later * np.arange(65)).sum()) / max(later.sum(), 1):.2f})")
    print(f"      forms: complement {st[176]}, direct {st[177]}, all passive {st[179]}, none passive {st[180]}")


if name.startswith("s_"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench_sparse
    gen, m, n, nnz_t, k, alg, desc = bench_sparse.SPARSE_WORKLOADS[name]
    A = bench_sparse.make_matrix(name)
    M = smallk_amd.SparseMatrix(A.data, A.indices, A.indptr, A.shape)
    W0 = smallk_amd.uniform_host(m, k, 43)
    H0 = smallk_amd.uniform_host(k, n, 44)
else:
    cfg = {"c2": (8192, 4096, 16, "uniform"), "c4s_uniform": (262144, 8192, 64, "uniform"), "c4s_planted": (262144, 8192, 64, "planted"),
           "mid32_uniform": (65536, 16384, 32, "uniform"), "mid32_planted": (65536, 16384, 32, "planted")}[name]
    m, n, k, data = cfg
    alg, desc = "BPP", f"dense {m}x{n} k={k} BPP fp32 {data}"
    M = smallk_amd.DenseMatrix(m, n)
    M.fill_uniform(42) if data == "uniform" else M.fill_planted(42, k, 0.7, 0.05)
    W0 = smallk_amd.uniform_host(m, k, 43)
    H0 = smallk_amd.uniform_host(k, n, 44)
print(f"== {name}: {desc}; BPP, {iters} iterations from the uniform start", flush=True)
s = smallk_amd.NmfSolver(M, smallk_amd.make_options(m, n, k, "BPP", min_iter=iters, max_iter=iters))
s.set_factors(W0, H0)
s.iterate(0); s.sync()
stats()
prevH = prevW = None
for it in range(1, iters + 1):
    s.iterate(1)
    assert s.sync() == 0
    st = stats()
    if it in CHECK:
        print(f" iteration {it}")
        show_stats("device counters (both solves of the iteration)", st)
        W, H = s.factors()
        mh, mw = masks(H), masks(np.ascontiguousarray(W.T))
        describe("H side", mh, prevH, k)
        describe("W side", mw, prevW, k)
        prevH, prevW = mh, mw
        sys.stdout.flush()
s.close()
M.close()
