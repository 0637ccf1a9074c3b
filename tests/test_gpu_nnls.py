"""NnlsBlockpivot on the device by itself (smk_nnls_blockpivot) against the oracle's restatement
(oracle/nmf_oracle.c:orc_nnls_blockpivot <- nnls.hpp:144-244), in the shapes the reference's own
tests/src/test_bpp.cpp:171-411 draws: random k, n == 1 every fifth run, random warm-start passive sets
(including empty and full ones), diagonally dominant Gram matrices, plus the cases that force the backup
rule / the pivot cap and a Gram matrix that is not positive definite.

Tolerance: the NNLS optimum is unique for an SPD Gram matrix; both sides work in fp64, so X and Y agree to
1e-9 of the largest entry (the reference's own threshold between its two solvers is 1e-10 on residuals)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def problem(rng, m, k, ncols, *, dominant=False, shift=True):
    W = rng.random((m, k))
    G = W.T @ W
    if dominant:                      # MakeDiagonallyDominant, tests/src/test_bpp.cpp
        G = G + np.diag(np.abs(G).sum(axis=1))
    B = W.T @ rng.random((m, ncols))
    if shift:
        B[:, ::3] -= 1.5 * np.abs(B[:, ::3]).mean()       # force active constraints
    return np.asfortranarray(G), np.asfortranarray(B)


def compare(gpu, G, B, X0, tol=1e-9):
    oko, Xo, Yo, _ = oracle.nnls_blockpivot(G, B, X0)
    okg, Xg, Yg = gpu.nnls_blockpivot(G, B, X0)
    assert okg == oko
    if not oko:
        return
    sx = max(np.abs(Xo).max(), 1e-300)
    sy = max(np.abs(Yo).max(), 1e-300)
    assert np.abs(Xg - Xo).max() <= tol * sx, np.abs(Xg - Xo).max() / sx
    assert np.abs(Yg - Yo).max() <= tol * max(sy, sx * np.abs(G).max()), np.abs(Yg - Yo).max()
    assert np.array_equal(Xg > 0, Xo > 0)                      # same passive sets
    # KKT, the reference's own acceptance test
    assert (Xg >= 0).all() and (Yg > -1e-9 * max(sy, 1.0)).all()
    assert np.abs(G @ Xg - B - Yg).max() < 1e-9 * max(np.abs(B).max(), 1.0)


@pytest.mark.parametrize("run", range(32))
def test_random_shapes_like_test_bpp(gpu, run):
    rng = np.random.default_rng(1000 + run)
    m = int(rng.integers(16, 768))
    n = int(rng.integers(64, 1024))
    k = int(rng.integers(4, min(m, n, 64) + 1))
    if run % 5 == 0:
        n = 1                                      # tests/src/test_bpp.cpp forces n == 1 every fifth run
    G, B = problem(rng, m, k, n, dominant=(run % 2 == 0))
    # random warm-start passive sets: mixed, plus some all-zero and all-positive columns
    X0 = rng.random((k, n)) * (rng.random((k, n)) < rng.random())
    if n > 3:
        X0[:, 0] = 0.0
        X0[:, 1] = 1.0
    compare(gpu, G, B, np.asfortranarray(X0))


@pytest.mark.parametrize("k", [33, 40, 48, 63, 64])
@pytest.mark.parametrize("fill", [0.0, 0.15, 0.5, 0.85, 1.0])
def test_k_above_32_every_passive_density(gpu, k, fill):
    """k in (32, 64] runs the inverse-based kernel: sparse and dense passive sets take its two forms
    (direct on G[F,F] when |F| < |Z|, complement on Ginv[Z,Z] otherwise), |F| = 0 and |Z| = 0 the shortcuts."""
    rng = np.random.default_rng(int(k * 10 + fill * 100))
    ncols = 300
    G, B = problem(rng, 4 * k + 5, k, ncols, shift=False)
    # steer the SOLUTION density: shift the right-hand side so that about `fill` of the entries end up passive
    B -= np.quantile(B, 1.0 - fill) if 0.0 < fill < 1.0 else (np.abs(B).max() * 2 if fill == 0.0 else 0.0)
    X0 = rng.random((k, ncols)) * (rng.random((k, ncols)) < fill)
    compare(gpu, G, B, np.asfortranarray(X0))


@pytest.mark.parametrize("k", [8, 16, 32, 64])
def test_ill_conditioned_gram_takes_the_slow_path_and_agrees(gpu, k):
    """Nearly collinear columns: cond(G) ~ 1e12.  For k = 64 the inversion guard rejects G and the masked
    Gauss-Jordan kernel runs; results still match the oracle at a tolerance that scales with cond."""
    rng = np.random.default_rng(k)
    W = rng.random((6 * k, k))
    W[:, 1] = W[:, 0] * (1 + 1e-6 * rng.random(6 * k))
    G = np.asfortranarray(W.T @ W)
    B = np.asfortranarray(W.T @ rng.random((6 * k, 50)))
    X0 = np.asfortranarray(rng.random((k, 50)))
    oko, Xo, Yo, _ = oracle.nnls_blockpivot(G, B, X0)
    okg, Xg, Yg = gpu.nnls_blockpivot(G, B, X0)
    assert okg == oko
    if oko:
        assert np.abs(G @ Xg - B - Yg).max() < 1e-6 * np.abs(B).max()
        assert np.abs(Xg - Xo).max() <= 1e-3 * np.abs(Xo).max()


@pytest.mark.parametrize("k", [65, 80, 100, 128])
@pytest.mark.parametrize("fill", [0.0, 0.2, 0.5, 0.8, 1.0])
def test_k_above_64_every_passive_density(gpu, k, fill):
    """k in (64, 128]: two components per lane, passive sets as two 64-bit words, compact solves up to 64 rows."""
    rng = np.random.default_rng(int(k * 10 + fill * 100))
    ncols = 200
    G, B = problem(rng, 4 * k + 5, k, ncols, shift=False)
    B -= np.quantile(B, 1.0 - fill) if 0.0 < fill < 1.0 else (np.abs(B).max() * 2 if fill == 0.0 else 0.0)
    X0 = rng.random((k, ncols)) * (rng.random((k, ncols)) < fill)
    compare(gpu, G, B, np.asfortranarray(X0))


@pytest.mark.parametrize("k", [129, 192, 250, 400])
@pytest.mark.parametrize("fill", [0.0, 0.3, 0.7, 1.0])
def test_k_above_128_every_passive_density(gpu, k, fill):
    """k in (128, 512]: a workgroup per column, Cholesky of the passive block in a global scratch panel (wide.hip)"""
    rng = np.random.default_rng(int(k * 10 + fill * 100))
    ncols = 60
    G, B = problem(rng, 4 * k + 5, k, ncols, shift=False)
    B -= np.quantile(B, 1.0 - fill) if 0.0 < fill < 1.0 else (np.abs(B).max() * 2 if fill == 0.0 else 0.0)
    X0 = rng.random((k, ncols)) * (rng.random((k, ncols)) < fill)
    compare(gpu, G, B, np.asfortranarray(X0))


@pytest.mark.parametrize("k,fill", [(513, 0.3), (600, 0.0), (600, 0.7), (800, 1.0), (1000, 0.4), (1024, 0.6)])
def test_k_above_512(gpu, k, fill):
    """k in (512, 1024]: the workgroup-per-column kernel with blocks of up to k / 2 rows (LDS up to 112 rows, else the
    workgroup's panel of global scratch).  Few columns: the oracle's scalar Cholesky is what takes the time."""
    rng = np.random.default_rng(int(k * 10 + fill * 100))
    ncols = 12
    G, B = problem(rng, 3 * k + 5, k, ncols, shift=False)
    B -= np.quantile(B, 1.0 - fill) if 0.0 < fill < 1.0 else (np.abs(B).max() * 2 if fill == 0.0 else 0.0)
    X0 = rng.random((k, ncols)) * (rng.random((k, ncols)) < fill)
    compare(gpu, G, B, np.asfortranarray(X0))


@pytest.mark.parametrize("k", [4, 16, 40, 64, 100, 130, 260, 700])
def test_not_positive_definite_is_failure(gpu, k):
    """Rank-one Gram matrix: the passive block is not SPD -> false (normal_eq.hpp:35-50)."""
    G = np.ones((k, k), order="F")
    B = np.ones((k, 6), order="F")
    oko, _, _, _ = oracle.nnls_blockpivot(G, B, np.ones((k, 6)))
    okg, _, _ = gpu.nnls_blockpivot(G, B, np.ones((k, 6)))
    assert not oko and not okg


@pytest.mark.parametrize("k", [12, 48, 64, 150])
def test_hard_problems_use_backup_rule(gpu, k):
    """Strongly correlated columns and alternating-sign right-hand sides make full exchanges cycle, so P
    runs out and the single-variable backup rule decides (src/nnls.cpp:52-70).  Results still agree."""
    rng = np.random.default_rng(77 + k)
    base = rng.random((3 * k, 1))
    W = base + 0.05 * rng.random((3 * k, k))
    G = np.asfortranarray(W.T @ W + 1e-3 * np.eye(k))
    B = np.asfortranarray(W.T @ (rng.random((3 * k, 200)) - 0.45))
    X0 = np.asfortranarray((rng.random((k, 200)) < 0.5) * 1.0)
    compare(gpu, G, B, X0, tol=1e-7)


def test_four_columns_per_wave_is_bit_identical_to_a_wave_per_column(tmp_path):
    """nnls_g16.hip (round 6: a column per 16-lane row, four per wave; columns of k > 32 whose exchange needs more than 16 rows go
    to the wave-per-column kernel through a work list) against nnls.hip's wave-per-column kernel on 63 problems: same operations in
    the same order, so the same bits -- and tools/nnls_g16_check.py's family covers both forms, t = 0 .. 32 and ragged column counts."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # SMK_NNLS_G16=2: the four-columns-per-wave kernel at k in (32, 64] as well (default: k <= 32 only)
    for tag, env in (("wave", {"SMK_NNLS_G16": "0"}), ("g16", {"SMK_NNLS_G16": "2"}), ("g16_shape1", {"SMK_NNLS_G16": "2", "SMK_NNLS_G16_SHAPE": "1"}), ("g16_shape0", {"SMK_NNLS_G16_SHAPE": "0"}),
                     ("default", {})):
        f = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "nnls_g16_check.py"), f], capture_output=True, text=True,
                           env=dict(os.environ, **env), cwd=root, timeout=600)
        assert r.returncode == 0 and "cases OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
        outs.append(np.load(f))
    ref = outs[0]
    for other in outs[1:]:
        assert sorted(ref.files) == sorted(other.files)
        for name in ref.files:
            assert np.array_equal(ref[name], other[name], equal_nan=True), name
