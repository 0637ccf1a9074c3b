"""Sparse input: MatrixMarket loader (CPU) and NmfSparse on the GPU against the dense oracle on
MakeDense(A) -- the reference's own sparse-vs-dense test shape (tests/src/test_dense_nmf.cpp:205-378,
threshold 1e-8 on ||dW||_F, ||dH||_F)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import oracle


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def random_sparse(m, n, density, seed):
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=density, random_state=rng, data_rvs=lambda s: rng.random(s) + 0.05, format="csc")
    # no empty rows / columns (they make W'W or HH' singular for BPP)
    A = A.tolil()
    for j in range(n):
        if A[:, j].nnz == 0:
            A[rng.integers(m), j] = 0.5
    for i in range(m):
        if A[i, :].nnz == 0:
            A[i, rng.integers(n)] = 0.5
    return A.tocsc()


def test_matrix_market_loader(tmp_path):
    import smallk_amd
    f = tmp_path / "g.mtx"
    f.write_text("%%MatrixMarket matrix coordinate real general\n% comment\n3 4 5\n1 1 1.5\n3 1 2.0\n2 3 -1.0\n1 4 4.0\n3 4 0.25\n")
    d, ri, co, shape = smallk_amd.load_matrix_market(f)
    A = sp.csc_matrix((d, ri, co), shape=shape).toarray()
    assert np.array_equal(A, np.array([[1.5, 0, 0, 4.0], [0, 0, -1.0, 0], [2.0, 0, 0, 0.25]]))
    g = tmp_path / "s.mtx"
    g.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 3\n1 1 2.0\n2 1 3.0\n3 2 4.0\n")
    d, ri, co, shape = smallk_amd.load_matrix_market(g)
    assert np.array_equal(sp.csc_matrix((d, ri, co), shape=shape).toarray(),
                          np.array([[2.0, 3.0, 0], [3.0, 0, 4.0], [0, 4.0, 0]]))
    h = tmp_path / "p.mtx"
    h.write_text("%%MatrixMarket matrix coordinate pattern general\n2 2 2\n1 2\n2 1\n")
    d, ri, co, shape = smallk_amd.load_matrix_market(h)
    assert np.array_equal(sp.csc_matrix((d, ri, co), shape=shape).toarray(), np.array([[0, 1.0], [1.0, 0]]))
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n")
    with pytest.raises(RuntimeError):
        smallk_amd.load_matrix_market(bad)          # fewer entries than announced
    arr = tmp_path / "arr.mtx"
    arr.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(RuntimeError):
        smallk_amd.load_matrix_market(arr)          # dense MatrixMarket is not a sparse input


@pytest.mark.gpu
@pytest.mark.parametrize("alg,m,n,k,density,iters", [
    ("MU", 300, 200, 8, 0.2, 20), ("BPP", 300, 200, 8, 0.2, 10), ("HALS", 300, 200, 8, 0.3, 10),
    ("RANK2", 300, 200, 2, 0.2, 20), ("BPP", 512, 400, 33, 0.1, 5), ("MU", 1000, 700, 64, 0.02, 10),
    ("BPP", 2000, 1500, 16, 0.05, 5), ("MU", 900, 700, 200, 0.3, 4), ("BPP", 900, 700, 150, 0.5, 3),
    # HALS is exercised only on the denser small case: on very sparse data rows of H die and the
    # reference update is discontinuous there (see tests/golden/make_golden.py; the reference's own
    # sparse-vs-dense test skips HALS, test_dense_nmf.cpp:263-266)
])
def test_sparse_matches_dense_oracle(gpu, alg, m, n, k, density, iters):
    A = random_sparse(m, n, density, seed=m + n + k)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    ref = oracle.nmf(A.toarray(), W0, H0, alg, min_iter=iters, max_iter=iters)
    got = gpu.nmf_sparse(A, W0, H0, alg, min_iter=iters, max_iter=iters)
    assert ref.result == 0 and got.result == 0 and got.iteration_count == iters
    eW, eH = np.linalg.norm(got.W - ref.W), np.linalg.norm(got.H - ref.H)
    assert eW < 1e-8 and eH < 1e-8, (eW, eH)


@pytest.mark.gpu
def test_sparse_solver_object_and_duplicates(gpu):
    """duplicate coordinates add up (SparseMatrix::Compress keeps them); solver object on a SparseMatrix"""
    m, n, k = 120, 90, 5
    A = random_sparse(m, n, 0.3, seed=7)
    # split every value into two entries at the same coordinate
    data = np.concatenate([A.data * 0.25, A.data * 0.75])
    indices = np.concatenate([A.indices, A.indices])
    counts = np.diff(A.indptr)
    cols = np.repeat(np.arange(n), counts)
    order = np.argsort(np.concatenate([cols, cols]), kind="stable")
    indptr = np.concatenate([[0], np.cumsum(2 * counts)])
    S = gpu.SparseMatrix(data[order], indices[order], indptr, (m, n))
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    s = gpu.NmfSolver(S, gpu.make_options(m, n, k, "BPP", min_iter=6, max_iter=6))
    s.set_factors(W0, H0)
    rc, it, _ = s.run()
    W, H = s.factors()
    ref = oracle.nmf(A.toarray(), W0, H0, "BPP", min_iter=6, max_iter=6)
    assert rc == 0 and it == 6
    assert rel(W, ref.W) < 1e-9 and rel(H, ref.H) < 1e-9


@pytest.mark.gpu
def test_smallkapi_sparse_inputs(gpu, tmp_path):
    """pysmallk flow with a MatrixMarket file and with CSC buffers (pysmallk/tests/smallkapi_inmem.py:57-94)"""
    m, n, k = 200, 150, 6
    A = random_sparse(m, n, 0.15, seed=11)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    ref = oracle.nmf(A.toarray(), W0, H0, "BPP", min_iter=1, max_iter=200, tol=0.01)
    l = gpu._lib.lib()
    dp = C.POINTER(C.c_double)
    fw, fh = str(tmp_path / "w0.csv"), str(tmp_path / "h0.csv")
    l.smk_write_csv(W0.ctypes.data_as(dp), m, m, k, fw.encode(), 17)
    l.smk_write_csv(H0.ctypes.data_as(dp), k, k, n, fh.encode(), 17)
    mtx = tmp_path / "a.mtx"
    coo = A.tocoo()
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{m} {n} {coo.nnz}\n")
        for r, c, v in zip(coo.row, coo.col, coo.data):
            f.write(f"{r + 1} {c + 1} {float(v)!r}\n")
    api = gpu.SmallkAPI()
    api.load_matrix(filepath=str(mtx))
    api.nmf(k, "BPP", infile_W=fw, infile_H=fh, min_iter=1, max_iter=200, tol=0.01, outdir=str(tmp_path))
    assert api.get_iteration_count() == ref.iteration_count
    assert rel(api.get_W(), ref.W) < 1e-8 and rel(api.get_H(), ref.H) < 1e-8
    api.load_matrix(height=m, width=n, nz=A.nnz, buffer=list(A.data), row_indices=list(A.indices),
                    col_offsets=list(A.indptr))
    api.nmf(k, "BPP", infile_W=fw, infile_H=fh, min_iter=1, max_iter=200, tol=0.01, outdir=str(tmp_path))
    assert rel(api.get_W(), ref.W) < 1e-8 and rel(api.get_H(), ref.H) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,lpc", [("2", None), ("4", "1"), ("8", "4"), ("1", "16")])
def test_rank2_gather_product_by_row_blocks_and_lanes(blocks, lpc, tmp_path):
    """The rank-2 gather product in its other shapes -- the gathered factor cut into 2 / 4 / 8 row blocks (spmm_blocked.hip:
    normally only above 6 MB of factor), 1 .. 16 lanes per column -- forced on a small matrix (the switches are read once per
    process, hence the child process): RANK2 on an uneven 700 x 450 sparse matrix against the oracle at 1e-8, both stopping
    rules, and the HierNMF2 tree of a planted matrix identical to the unblocked one."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, scipy.sparse as sp
import oracle, smallk_amd
from test_sparse import random_sparse, rel
from hier_cases import planted
smallk_amd.initialize(0)
A = random_sparse(700, 450, 0.05, 5)
W0 = oracle.fill_uniform(700, 2, 3); H0 = oracle.fill_uniform(2, 450, 4)
for kw in (dict(min_iter=12, max_iter=12), dict(min_iter=2, max_iter=400, tol=1e-3)):
    ref = oracle.nmf_sparse(A, W0, H0, "RANK2", **kw)
    got = smallk_amd.nmf_sparse(A, W0, H0, "RANK2", **kw)
    assert got.result == ref.result == 0 and got.iteration_count == ref.iteration_count, (got.result, got.iteration_count, ref.iteration_count)
    assert rel(got.W, ref.W) < 1e-8 and rel(got.H, ref.H) < 1e-8, (rel(got.W, ref.W), rel(got.H, ref.H))
P, _ = planted(300, 400, 6, 2, sparse=True)
t = smallk_amd.hier_nmf2(P, 6, seed=2)
print("ASSIGN", ",".join(str(int(x)) for x in t.get_assignments()))
""" % (root, os.path.join(root, "tests"))
    outs = []
    for env in ({"SMK_SPMM_BLOCKS": blocks, **({"SMK_SPMM2_LPC": lpc, "SMK_SPMM_BLOCKED_LPC": lpc if lpc in ("1", "2", "4", "8") else "8"} if lpc else {})},
                {"SMK_SPMM_BLOCKS": "1"}):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("ASSIGN")][0])
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_resident_rank2_kernel_matches_the_launch_per_kernel_loop():
    import os
    """RANK2 on sparse A runs as ONE resident launch (rank2_persist.hip: the NmfSolve<> loop, stopping rule and
    per-iteration normalisation inside the kernel, two grid-wide barriers per iteration).  tools/r2_persist_check.py runs
    the same problems -- converging with a tolerance, min_iter = max_iter, rectangular, one and two iterations -- in two
    processes, SMK_R2_PERSIST=0 and on, and asks for equal result codes and iteration counts and factors equal to 1e-9
    (they differ by summation order: 1e-16 measured)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "tools/r2_persist_check.py", "quick"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert "fell back" not in r.stdout


@pytest.mark.gpu
def test_resident_rank2_kernel_falls_back_when_it_cannot_synchronise(tmp_path):
    """If the resident kernel reports that its workgroups could not all synchronise (a bounded wait expired: somebody else
    holds the CUs), the run continues on the launch-per-kernel loop from the state solver.Init left and the handle stays on
    that path.  TEST HOOK SMK_R2P_TEST_ABORT=1 reports that outcome without launching; result = the oracle's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np, scipy.sparse as sp\n"
            "sys.path.insert(0, '.'); import oracle, smallk_amd as g\n"
            "g.initialize(0)\n"
            "rng = np.random.default_rng(3)\n"
            "A = sp.random(900, 700, density=0.02, random_state=rng, data_rvs=lambda s: rng.random(s) + 0.1, format='csc') + 0.01 * sp.eye(900, 700, format='csc')\n"
            "W0 = oracle.fill_uniform(900, 2, 5); H0 = oracle.fill_uniform(2, 700, 6)\n"
            "r = g.nmf_sparse(A.tocsc(), W0, H0, 'RANK2', min_iter=3, max_iter=300, tol=1e-3)\n"
            "ref = oracle.nmf(np.asfortranarray(A.toarray()), W0, H0, 'RANK2', min_iter=3, max_iter=300, tol=1e-3)\n"
            "print('RES', r.result, r.iteration_count, ref.iteration_count, float(np.linalg.norm(r.W - ref.W) / np.linalg.norm(ref.W)))\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, SMK_R2P_TEST_ABORT="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    res = [l for l in out.stdout.splitlines() if l.startswith("RES")][0].split()
    assert res[1] == "0" and res[2] == res[3] and float(res[4]) < 1e-8, res
    assert "continues on the launch-per-kernel path" in out.stderr
