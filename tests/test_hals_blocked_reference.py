"""The blocked form of the HALS W sweep (smallk_amd/csrc/wide.hip: launch_hals_w_update_blocked) restated in numpy next to the
plain sweep (nmf_solver_hals.hpp:66-117 as oracle/nmf_oracle.c restates it): the dots w_i . G[:, c] of a block of 16 columns
come from ONE product with the W the block started from, corrected by what changed since -- the normalisation of the column
before the block and the updates + normalisations of the block's earlier columns.  Same numbers up to rounding; CPU only."""
import numpy as np
import pytest


def plain_sweep(W, AHt, G):
    W = W.copy()
    m, k = W.shape
    for c in range(k):
        t = W[:, c] + (AHt[:, c] - W @ G[:, c]) / G[c, c]
        t = np.where(np.isnan(t) | (t < 0), 0.0, t)
        if not t.any():
            t[:] = np.finfo(np.float64).eps
        W[:, c] = t / np.linalg.norm(t)
    return W


def blocked_sweep(W, AHt, G, nb=16):
    W = W.copy()
    m, k = W.shape
    pending = None                       # (column, scale or fill) whose normalisation the next launch applies
    for c0 in range(0, k, nb):
        c1 = min(c0 + nb, k)
        Y = W @ G[:, c0:c1]              # the block's product: sees column c0 - 1 still unnormalised
        delta = np.zeros((m, 1 + nb))    # slot 0: column c0 - 1; slot 1 + q: column c0 + q
        for c in range(c0, c1):
            nq = c - c0
            if pending is not None:      # launch c first finishes column c - 1
                p, scale, fill = pending
                new = np.full(m, fill) if fill is not None else W[:, p] * scale
                slot = 0 if p < c0 else 1 + p - c0
                delta[:, slot] += new - W[:, p]
                W[:, p] = new
            dot = Y[:, nq].copy()
            if c0 > 0:
                dot += delta[:, 0] * G[c0 - 1, c]
            for q in range(nq):
                dot += delta[:, 1 + q] * G[c0 + q, c]
            t = W[:, c] + (AHt[:, c] - dot) / G[c, c]
            t = np.where(np.isnan(t) | (t < 0), 0.0, t)
            delta[:, 1 + nq] = t - W[:, c]
            W[:, c] = t
            if not t.any():
                eps = np.finfo(np.float64).eps
                pending = (c, None, eps / np.sqrt(m * eps * eps))
            else:
                pending = (c, 1.0 / np.linalg.norm(t), None)
    p, scale, fill = pending             # the launch after the last column
    W[:, p] = np.full(m, fill) if fill is not None else W[:, p] * scale
    return W


@pytest.mark.parametrize("m,k,seed", [(50, 5, 0), (200, 16, 1), (300, 17, 2), (257, 70, 3), (120, 100, 4)])
def test_blocked_w_sweep_equals_plain_sweep(m, k, seed):
    rng = np.random.default_rng(seed)
    n = k + 30
    H = rng.random((k, n)) * (rng.random((k, n)) > 0.3)
    A = rng.random((m, n))
    W = rng.random((m, k)) * (rng.random((m, k)) > 0.2)
    if k > 6:
        W[:, 3] = 0.0                    # a column that the update may leave all zero ...
        A[:, :] = np.maximum(A - 0.2, 0.0)
    G, AHt = H @ H.T, A @ H.T
    a, b = plain_sweep(W, AHt, G), blocked_sweep(W, AHt, G)
    assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max())
    assert np.allclose(np.linalg.norm(b, axis=0), 1.0)


def test_all_zero_column_takes_the_eps_guard():
    m, k = 40, 20
    rng = np.random.default_rng(9)
    H = rng.random((k, 50))
    G = H @ H.T
    W = rng.random((m, k))
    AHt = W @ G
    AHt[:, 7] = -1e6                     # forces column 7 to clamp to zero everywhere
    a, b = plain_sweep(W, AHt, G), blocked_sweep(W, AHt, G)
    assert np.allclose(a[:, 7], 1.0 / np.sqrt(m)) and np.abs(a - b).max() <= 1e-12
