#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- small W,H fixtures for the dense NMF path.

PROVENANCE (read before trusting these numbers): the reference cannot be built
or imported in this image (Elemental is an empty submodule; pysmallk is py2
Cython over the same C++), and the reference tree holds no golden vectors for
this path.  These fixtures are therefore produced by an INDEPENDENT numpy/scipy
restatement written straight from the algorithm descriptions in SURVEY.md 8(a):

  * MU   : nmf_solver_mu.hpp:121-164        (plain numpy)
  * HALS : nmf_solver_hals.hpp:26-117,166-199 (plain numpy, Gauss-Seidel loops)
  * BPP  : nmf_solver_bpp.hpp:342-377 with each NNLS sub-problem solved by
           scipy.optimize.nnls (Lawson-Hanson active set) on the Cholesky factor
           of the Gram matrix -- a *different* algorithm from block principal
           pivoting; it agrees because the NNLS optimum is unique for SPD Gram.
  * driver: nmf_solve_generic.hpp:34-140 (fixed iteration count, final
           NormalizeAndScale normalize.hpp:118-140)

They pin the C oracle (oracle/nmf_oracle.c) against a second, independent
implementation; they do not pin either against the reference binary.

Inputs are not stored: A, W0, H0 come from the counter-based generator
(`uniform()` below, identical to oracle.fill_uniform / the device generator)
with seeds 42/43/44.

Usage:  python tests/golden/make_golden.py      (writes next to this file)
"""
import os
import sys

import numpy as np
from scipy.optimize import nnls as scipy_nnls

HERE = os.path.dirname(os.path.abspath(__file__))

MASK64 = (1 << 64) - 1


def _mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def uniform(rows, cols, seed, quant=0):
    """uniform [0,1) with 24 random bits; element (r,c) hashed from c*rows + r."""
    idx = np.arange(rows * cols, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + idx
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    f = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    if quant == 1:
        f = bf16_round(f)
    return f.astype(np.float64).reshape((rows, cols), order="F")


def bf16_round(f32):
    b = f32.astype(np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def normalize_and_scale(W, H):
    nrm = np.sqrt((W * W).sum(axis=0))
    return W / nrm, H * nrm[:, None]


def mu(A, W, H, iters):
    eps = 1e-13
    WtA = W.T @ A
    WtW = W.T @ W
    for _ in range(iters):
        H = H * (WtA / (WtW @ H + eps))
        HHt = H @ H.T
        AHt = A @ H.T
        W = W * (AHt / (W @ HHt + eps))
        WtA = W.T @ A
        WtW = W.T @ W
    return W, H


def hals(A, W, H, iters):
    k = W.shape[1]
    W = W.copy()
    H = H.copy()
    HHt = H @ H.T
    AHt = A @ H.T
    for _ in range(iters):
        for c in range(k):
            w = W[:, c] + (AHt[:, c] - W @ HHt[:, c]) / HHt[c, c]
            w = np.where(np.isnan(w) | (w < 0), 0.0, w)
            if not w.any():
                w[:] = np.finfo(np.float64).eps
            W[:, c] = w / np.linalg.norm(w)
        WtW = W.T @ W
        WtA = W.T @ A
        for r in range(k):
            h = H[r, :] + (WtA[r, :] - WtW[r, :] @ H) / WtW[r, r]
            H[r, :] = np.where(np.isnan(h) | (h < 0), 0.0, h)
        HHt = H @ H.T
        AHt = A @ H.T
    return W, H


def nnls_gram(G, B):
    """argmin_X>=0 0.5 x'Gx - b'x per column, via Lawson-Hanson on the Cholesky factor."""
    R = np.linalg.cholesky(G).T                       # G = R'R
    C = np.linalg.solve(R.T, B)                       # R'c = b  ->  ||Rx - c||^2
    X = np.empty_like(B)
    for j in range(B.shape[1]):
        X[:, j], _ = scipy_nnls(R, C[:, j], maxiter=50 * G.shape[0])
    return X


def bpp(A, W, H, iters):
    for _ in range(iters):
        H = nnls_gram(W.T @ W, W.T @ A)
        W = nnls_gram(H @ H.T, H @ A.T).T
    return W, H


# (m, n, k, planted) -- SURVEY 8(c): C1, tail-word bitmask (33), 2-word (64), small-n, k==n.
# The two high-rank cases use a planted low-rank A: on pure uniform noise at
# k=33/64 some rows of H die (become exactly zero) within two HALS sweeps, and
# the reference's HALS update (0/0 -> NaN -> 0 -> epsilon column,
# nmf_solver_hals.hpp:86-111) is discontinuous there, so two correct
# implementations that differ by one ulp diverge by O(1).  The reference's own
# test skips HALS for this reason (tests/src/test_dense_nmf.cpp:263-266).
CASES = [
    (96, 64, 5, False), (512, 256, 8, False), (300, 200, 33, True), (256, 192, 64, True),
    (64, 16, 4, False), (40, 8, 8, False),
]


def make_A(m, n, k, planted, quant):
    """Test matrix, rounded to what the device stores (quant 0: fp32, 1: bf16)."""
    if not planted:
        return uniform(m, n, 42, quant)
    Ws = uniform(m, k, 45)
    Hs = uniform(k, n, 46)
    Ws = np.where(Ws > 0.7, Ws, 0.0)
    Hs = np.where(Hs > 0.7, Hs, 0.0)
    A = (Ws @ Hs + 0.05 * uniform(m, n, 42)).astype(np.float32)
    if quant == 1:
        A = bf16_round(A)
    return np.asfortranarray(A.astype(np.float64))
def rank2(A, W, H, iters):
    """nmf_solver_rank2.hpp:353-455 with the 2x2 systems solved by numpy (the reference uses one
    fast Givens rotation) and the optimal-active-set rule (:216-318); per-iteration normalisation."""
    def solve_side(G, B):
        X = np.linalg.solve(G, B)
        v1 = B[0] / G[0, 0]
        v2 = B[1] / G[1, 1]
        c = v1 * np.sqrt(G[0, 0]) >= v2 * np.sqrt(G[1, 1])
        alt = np.vstack([np.where(c, v1, 0.0), np.where(c, 0.0, v2)])
        bad = (X[0] <= 0) | (X[1] <= 0)
        return np.where(bad, alt, X)
    for _ in range(iters):
        H = solve_side(W.T @ W, W.T @ A)
        W = solve_side(H @ H.T, (A @ H.T).T).T
        nrm = np.linalg.norm(W, axis=0)
        W = W / nrm
        H = H * nrm[:, None]
    return W, H


RANK2_CASES = [(300, 200), (512, 256), (64, 16)]
ITERS = (1, 5, 20)
ALGS = {"MU": mu, "HALS": hals, "BPP": bpp}


def main():
    out = {}
    for (m, n, k, planted) in CASES:
        for quant in (0, 1):
            A = make_A(m, n, k, planted, quant)
            W0 = uniform(m, k, 43)
            H0 = uniform(k, n, 44)
            for name, fn in ALGS.items():
                for it in ITERS:
                    if quant == 1 and it != 5:
                        continue
                    W, H = fn(A, W0.copy(), H0.copy(), it)
                    W, H = normalize_and_scale(W, H)
                    key = f"{name}_{m}x{n}_k{k}_it{it}_q{quant}"
                    out[key + "_W"] = W
                    out[key + "_H"] = H
                    print(key, float(np.linalg.norm(A - W @ H) / np.linalg.norm(A)))
    for (m, n) in RANK2_CASES:
        for quant in (0, 1):
            A = uniform(m, n, 42, quant)
            W0 = uniform(m, 2, 43)
            H0 = uniform(2, n, 44)
            for it in ITERS:
                if quant == 1 and it != 5:
                    continue
                W, H = rank2(A, W0.copy(), H0.copy(), it)
                W, H = normalize_and_scale(W, H)
                key = f"RANK2_{m}x{n}_k2_it{it}_q{quant}"
                out[key + "_W"] = W
                out[key + "_H"] = H
                print(key, float(np.linalg.norm(A - W @ H) / np.linalg.norm(A)))
    np.savez_compressed(os.path.join(HERE, "nmf_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    sys.exit(main())


def make_planted_fixture(path):
    """tests/golden/planted_generator.npz: a 48 x 40 block of the planted-data generator (orc_fill_planted, seed 2024, rank 19,
    threshold 0.7, noise 0.05) in fp32 and bf16 storage -- pins the generator's bits (tests/test_round5_cpu.py).  Needs the
    oracle library: python -c "import sys; sys.path.insert(0, 'tests/golden'); import make_golden as m; m.make_planted_fixture('tests/golden/planted_generator.npz')" """
    import oracle
    np.savez_compressed(path, f32=oracle.fill_planted(48, 40, 2024, 19), bf16=oracle.fill_planted(48, 40, 2024, 19, quant=1),
                        params=np.array([48, 40, 2024, 19]))
