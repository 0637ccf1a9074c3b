"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors, against
scipy's NNLS, and against the reference's own property tests."""
import numpy as np
import pytest

import oracle
import make_golden as mg


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_generator_matches_golden_script():
    for quant in (0, 1):
        a = mg.uniform(37, 11, 42, quant)
        b = oracle.fill_uniform(37, 11, 42, quant=quant)
        assert np.array_equal(a, b)
    # sub-block reproducibility (column shards see the same values)
    full = oracle.fill_uniform(64, 48, 7)
    part = oracle.fill_uniform(64, 16, 7, c0=16, gheight=64)
    assert np.array_equal(full[:, 16:32], part)


CASES = [(m, n, k, pl, q, alg, it)
         for (m, n, k, pl) in mg.CASES for q in (0, 1) for alg in ("MU", "HALS", "BPP")
         for it in ((1, 5, 20) if q == 0 else (5,))]


@pytest.mark.parametrize("m,n,k,planted,quant,alg,iters", CASES)
def test_oracle_matches_independent_restatement(golden, m, n, k, planted, quant, alg, iters):
    """oracle (C, block principal pivoting) == numpy/scipy restatement (Lawson-Hanson NNLS)."""
    A = mg.make_A(m, n, k, planted, quant)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    r = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters)
    key = f"{alg}_{m}x{n}_k{k}_it{iters}_q{quant}"
    assert r.result == oracle.OK
    assert r.iteration_count == iters
    assert rel(r.W, golden[key + "_W"]) < 1e-10
    assert rel(r.H, golden[key + "_H"]) < 1e-10
    # final NormalizeAndScale: unit column norms
    assert np.allclose(np.linalg.norm(r.W, axis=0), 1.0, atol=1e-12)


@pytest.mark.parametrize("k,ncols", [(16, 200), (33, 300), (64, 128), (8, 1), (1, 50)])
def test_nnls_kkt_and_scipy(k, ncols):
    """NnlsBlockpivot: KKT conditions (reference tests/src/test_bpp.cpp thresholds 1e-10) and
    agreement with scipy's active-set NNLS (unique optimum for SPD Gram)."""
    from scipy.optimize import nnls
    rng = np.random.default_rng(k * 1000 + ncols)
    Wm = rng.random((4 * k + 5, k))
    G = Wm.T @ Wm
    B = Wm.T @ rng.random((4 * k + 5, ncols))
    B[:, ::3] -= 1.5 * np.abs(B[:, ::3]).mean()       # force active constraints
    ok, X, Y, piv = oracle.nnls_blockpivot(G, B, rng.random((k, ncols)))
    assert ok
    assert (X >= 0).all()
    assert np.abs(G @ X - B - Y).max() < 1e-9
    assert (Y > -1e-9).all()
    assert np.abs(X * Y).max() < 1e-8
    R = np.linalg.cholesky(G).T
    C = np.linalg.solve(R.T, B)
    for j in range(0, ncols, max(1, ncols // 10)):
        xs, _ = nnls(R, C[:, j], maxiter=100 * k)
        assert np.abs(xs - X[:, j]).max() < 1e-8


def test_nnls_failure_on_rank_deficient_gram():
    """non-SPD sub-problem -> false (normal_eq.hpp:35-50) -> Result::FAILURE from Nmf()."""
    k, n = 4, 6
    G = np.ones((k, k))                 # rank 1
    B = np.ones((k, n))
    ok, X, Y, _ = oracle.nnls_blockpivot(G, B, np.ones((k, n)))
    assert not ok
    A = np.ones((12, 6))
    r = oracle.nmf(A, np.ones((12, 3)), np.ones((3, 6)), "BPP", min_iter=1, max_iter=2)
    assert r.result == oracle.FAILURE


def test_driver_stopping_rules():
    """NmfSolve<> control flow: min_iter branch, tolerance, iteration_count semantics
    (nmf_solve_generic.hpp:67-139)."""
    m, n, k = 96, 64, 5
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    r = oracle.nmf(A, W0, H0, "BPP", min_iter=3, max_iter=200, tol=0.05)
    assert r.result == oracle.OK and 3 <= r.iteration_count < 200
    # metric recorded at iter 0 (==1 for PG ratio) and from min_iter on
    assert r.metrics[0] == 1.0 and np.isnan(r.metrics[1]) and np.isnan(r.metrics[2])
    assert r.metrics[r.iteration_count] <= 0.05
    # reaching max_iter is success; iteration_count == max_iter
    r2 = oracle.nmf(A, W0, H0, "MU", min_iter=2, max_iter=4, tol=1e-9)
    assert r2.result == oracle.OK and r2.iteration_count == 4
    # invalid options (nmf_options.cpp:23-112)
    assert oracle.nmf(A, W0, H0, "MU", tol=1.5).result == oracle.BAD_PARAM
    assert oracle.nmf(A[:, :3], W0, H0[:, :3], "MU").result == oracle.BAD_PARAM     # k > n


def test_hals_residual_decreases():
    m, n, k = 300, 200, 33
    A = mg.make_A(m, n, k, True, 0)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    errs = []
    for it in (1, 5, 20):
        r = oracle.nmf(A, W0, H0, "HALS", min_iter=it, max_iter=it)
        errs.append(np.linalg.norm(A - r.W @ r.H))
    assert errs[0] > errs[1] > errs[2]


R2CASES = [(m, n, q, it) for (m, n) in mg.RANK2_CASES for q in (0, 1) for it in ((1, 5, 20) if q == 0 else (5,))]


@pytest.mark.parametrize("m,n,quant,iters", R2CASES)
def test_rank2_oracle_matches_independent_restatement(golden, m, n, quant, iters):
    """Solver_Generic_Rank2 (nmf_solver_rank2.hpp:323-461): fast-Givens oracle == numpy 2x2 solves"""
    A = mg.uniform(m, n, 42, quant)
    W0 = oracle.fill_uniform(m, 2, 43)
    H0 = oracle.fill_uniform(2, n, 44)
    r = oracle.nmf(A, W0, H0, "RANK2", min_iter=iters, max_iter=iters)
    key = f"RANK2_{m}x{n}_k2_it{iters}_q{quant}"
    assert r.result == oracle.OK and r.iteration_count == iters
    assert rel(r.W, golden[key + "_W"]) < 1e-10 and rel(r.H, golden[key + "_H"]) < 1e-10


def test_rank2_requires_k2_and_solves_2x2_exactly():
    A = oracle.fill_uniform(50, 40, 42)
    assert oracle.nmf(A, np.ones((50, 3)), np.ones((3, 40)), "RANK2").result == oracle.BAD_PARAM
    # residual of the 2x2 solves (reference tests/src/test_rank2_system_solve.cpp: ||AX-B|| < 1e-10):
    # one iteration from a W0 whose unconstrained solution is positive gives H = (W'W)^-1 W'A
    rng = np.random.default_rng(3)
    W0 = rng.random((50, 2)) + 0.5
    Hs = rng.random((2, 40)) + 0.5
    A2 = W0 @ Hs
    r = oracle.nmf(A2, W0, np.ones((2, 40)), "RANK2", min_iter=1, max_iter=1, normalize=False)
    assert np.linalg.norm(A2 - r.W @ r.H) / np.linalg.norm(A2) < 1e-10


@pytest.mark.parametrize("alg,k", [("MU", 5), ("HALS", 4), ("BPP", 6), ("RANK2", 2)])
def test_sparse_driver_equals_dense_driver(alg, k):
    """orc_nmf_sparse (NmfSparse: products over the stored entries) against orc_nmf on the densified matrix:
    the reference's own test (tests/src/test_dense_nmf.cpp:205-378) demands 1e-8 between the two."""
    import scipy.sparse as sp
    rng = np.random.default_rng(k)
    m, n = 120, 90
    A = sp.random(m, n, density=0.15, random_state=3, format="csc", data_rvs=lambda s: rng.random(s) + 0.1)
    W0, H0 = oracle.fill_uniform(m, k, 5), oracle.fill_uniform(k, n, 6)
    a = oracle.nmf_sparse(A, W0, H0, alg, min_iter=8, max_iter=8, tol=1e-12)
    b = oracle.nmf(A.toarray(order="F"), W0, H0, alg, min_iter=8, max_iter=8, tol=1e-12)
    assert a.result == b.result == 0 and a.iteration_count == b.iteration_count
    assert np.abs(a.W - b.W).max() < 1e-10 * np.abs(b.W).max()
    assert np.abs(a.H - b.H).max() < 1e-10 * np.abs(b.H).max()
