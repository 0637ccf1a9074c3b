"""The algebra the block-pivoting kernels for k > 32 rest on (smallk_amd/csrc/nnls.hip, wide.hip), checked in
numpy: (1) a passive set F can be solved through its complement Z and the inverse of the Gram matrix -- the form the kernels
take whenever |Z| < |F|; (2) the tiled, right-looking Cholesky with look-ahead (tiles_cholesky<NW > 1>) is a Cholesky
factorisation; (3) the tile offsets enumerate the lower triangle without gaps.  CPU only."""
import numpy as np
import pytest


def spd(k, seed):
    rng = np.random.default_rng(seed)
    X = rng.random((3 * k + 5, k))
    return X.T @ X


@pytest.mark.parametrize("k,seed", [(40, 0), (64, 1), (100, 2), (200, 3)])
def test_complement_form_equals_direct_form(k, seed):
    """x_F = G_FF^-1 r_F, x_Z = 0, y = G x - r   ==   y_Z = -(Ginv_ZZ)^-1 v_Z, x = v + Ginv[:, Z] y_Z   with v = Ginv r"""
    rng = np.random.default_rng(100 + seed)
    G = spd(k, seed)
    r = rng.standard_normal(k) * np.abs(G).mean()
    passive = rng.random(k) < 0.7
    F, Z = np.flatnonzero(passive), np.flatnonzero(~passive)
    x = np.zeros(k)
    x[F] = np.linalg.solve(G[np.ix_(F, F)], r[F])
    y = G @ x - r
    Ginv = np.linalg.inv(G)
    v = Ginv @ r
    yz = -np.linalg.solve(Ginv[np.ix_(Z, Z)], v[Z])
    x2 = v + Ginv[:, Z] @ yz
    scale = np.abs(x).max()
    assert np.abs(x2[Z]).max() <= 1e-9 * scale               # x_Z = 0 comes out of the formula
    assert np.abs(x2[F] - x[F]).max() <= 1e-9 * scale
    assert np.abs(yz - y[Z]).max() <= 1e-9 * max(np.abs(y).max(), 1.0)
    assert np.abs(y[F]).max() <= 1e-9 * max(np.abs(y).max(), 1.0)   # y_F = 0


def tile_off(I, L):
    return (I * (I + 1) // 2 + L)


def test_tile_offsets_enumerate_the_lower_triangle():
    tp = 9
    seen = sorted(tile_off(I, L) for I in range(tp) for L in range(I + 1))
    assert seen == list(range(tp * (tp + 1) // 2))


def tiled_cholesky_lookahead(M, nb=16):
    """The schedule of tiles_cholesky<NW > 1>: diagonal tile 0; then per tile column J: the panel rows below it, the tiles of
    column J + 1, and only then (beside the next diagonal tile) the rest of the trailing triangle.  Padding rows / columns are
    those of the identity.  Returns L."""
    t = M.shape[0]
    tp = -(-t // nb)
    T = np.eye(tp * nb)
    T[:t, :t] = M
    blk = lambda I, L: (slice(nb * I, nb * I + nb), slice(nb * L, nb * L + nb))

    def factor_diag(J):
        T[blk(J, J)] = np.linalg.cholesky(T[blk(J, J)])
    factor_diag(0)
    for J in range(tp - 1):
        Ljj = T[blk(J, J)]
        for I in range(J + 1, tp):                           # panel: rows <- row L_JJ^-T
            T[blk(I, J)] = np.linalg.solve(Ljj, T[blk(I, J)].T).T
        for I in range(J + 1, tp):                           # tile column J + 1 first
            T[blk(I, J + 1)] -= T[blk(I, J)] @ T[blk(J + 1, J)].T
        factor_diag(J + 1)                                   # (beside ...)
        for I in range(J + 2, tp):                           # ... the rest of the trailing triangle
            for L in range(J + 2, I + 1):
                T[blk(I, L)] -= T[blk(I, J)] @ T[blk(L, J)].T
    return np.tril(T)[:t, :t]


@pytest.mark.parametrize("t,seed", [(7, 0), (16, 1), (17, 2), (57, 3), (96, 4), (130, 5)])
def test_tiled_cholesky_with_lookahead(t, seed):
    M = spd(t, 10 + seed)
    L = tiled_cholesky_lookahead(M)
    assert np.abs(L @ L.T - M).max() <= 1e-10 * np.abs(M).max()
    assert np.abs(L - np.linalg.cholesky(M)).max() <= 1e-9 * np.abs(M).max() ** 0.5
