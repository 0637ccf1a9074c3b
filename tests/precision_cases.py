"""Fast product form against the accurate one on the SAME resident A (helper shared by tests/test_gpu_precision.py and
tools/fast_vs_accurate.py).

The default product form of a dense solver (fp16 two-term / bf16x3 operands on the matrix cores) and the
accurate form (SMK_NSPLIT=8: the stored A against the fp64 factor, fp64 accumulation) start from the same factors and run
the same number of iterations; the factors are compared at checkpoints.  Together with one accurate-form iteration checked
against the oracle on sampled rows / columns this closes the chain oracle <-> accurate <-> fast at sizes the oracle cannot
hold (reference loop: common/include/nmf_solver_bpp.hpp:342-377, nnls.hpp:144-244)."""
import os

import numpy as np

CHECKPOINTS = (1, 5, 10, 25, 50)


def fro(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


class _Env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def make_matrix(m, n, data, seed, kstar, storage="f32"):
    from smallk_amd import DenseMatrix
    A = DenseMatrix(m, n, storage=storage)
    if data == "uniform":
        A.fill_uniform(seed)
    else:
        A.fill_planted(seed, kstar, 0.7, 0.05)
    return A


def fast_vs_accurate(A, k, alg, seeds, checkpoints=CHECKPOINTS, fast_env=None, h0_scale=None):
    """[(iteration, relF(W_fast, W_acc), relF(H_fast, H_acc))], the forms, and the accurate factors at the last checkpoint.
    h0_scale: start from uniform W0 and h0_scale * uniform H0 (HALS: an unscaled uniform start clamps all of W to zero in its
    first sweep, the run would then iterate on the epsilon guard columns -- bench.py starts HALS from 2 / k as well)."""
    from smallk_amd import NmfSolver, make_options, uniform_host
    m, n = A.height, A.width_global
    opts = make_options(m, n, k, alg, normalize=False)
    with _Env(SMK_NSPLIT=fast_env, SMK_BPP_SMALL_ACCURATE="0"):
        fast = NmfSolver(A, opts)
    with _Env(SMK_NSPLIT="8"):
        acc = NmfSolver(A, opts)
    forms = (fast.product_form()[0], acc.product_form()[0])
    if h0_scale is not None:
        W0, H0 = uniform_host(m, k, seeds[0]), uniform_host(k, n, seeds[1]) * h0_scale
    for s in (fast, acc):
        if h0_scale is not None:
            s.set_factors(W0, H0)
        else:
            s.set_factors_uniform(seeds[0], seeds[1])
    rows, done = [], 0
    Wa = Ha = None
    for cp in checkpoints:
        for s in (fast, acc):
            s.iterate(cp - done)
            rc = s.sync()
            assert rc == 0, (cp, rc)
        done = cp
        Wf, Hf = fast.factors(normalize=False)
        Wa, Ha = acc.factors(normalize=False)
        assert np.isfinite(Wf).all() and np.isfinite(Hf).all() and np.isfinite(Wa).all() and np.isfinite(Ha).all()
        rows.append((cp, fro(Wf, Wa), fro(Hf, Ha)))
    fast.close()
    acc.close()
    return rows, forms, (Wa, Ha)


def oracle_block(oracle, data, m, seed, kstar, quant, *, cols=None, rows=None, n=None):
    """Sampled columns (m x len(cols)) or rows (len(rows) x n) of the generated A, from the oracle's twin of the generator."""
    if cols is not None:
        gen = (lambda c: oracle.fill_uniform(m, 1, seed, quant=quant, c0=int(c), gheight=m)) if data == "uniform" else \
              (lambda c: oracle.fill_planted(m, 1, seed, kstar, quant=quant, c0=int(c), gheight=m))
        return np.asfortranarray(np.concatenate([gen(c) for c in cols], axis=1))
    gen = (lambda r: oracle.fill_uniform(1, n, seed, quant=quant, r0=int(r), gheight=m)) if data == "uniform" else \
          (lambda r: oracle.fill_planted(1, n, seed, kstar, quant=quant, r0=int(r), gheight=m))
    return np.asfortranarray(np.concatenate([gen(r) for r in rows], axis=0))


def accurate_iteration_vs_oracle(oracle, A, k, alg, data, seed, kstar, seeds, nsample=None):
    """One accurate-form iteration at full size against the oracle on sampled columns (H update) and rows (W update):
    returns (max relative error over the sampled H columns, same for W rows, passive sets equal)."""
    from smallk_amd import NmfSolver, make_options, uniform_host
    m, n = A.height, A.width_global
    nsample = nsample or k + 8
    with _Env(SMK_NSPLIT="8"):
        s = NmfSolver(A, make_options(m, n, k, alg, normalize=False))
    W0, H0 = uniform_host(m, k, seeds[0]), uniform_host(k, n, seeds[1])
    s.set_factors(W0, H0)
    s.iterate(1)
    assert s.sync() == 0
    W1, H1 = s.factors(normalize=False)
    s.close()
    rng = np.random.default_rng(5)
    cols = np.sort(rng.choice(n, size=nsample, replace=False))
    Ac = oracle_block(oracle, data, m, seed, kstar, 0, cols=cols)
    ref = oracle.nmf(Ac, W0, H0[:, cols], alg, min_iter=1, max_iter=1, normalize=False)
    assert ref.result == 0
    relerr = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
    eH = relerr(H1[:, cols], ref.H)
    same = bool(np.array_equal(H1[:, cols] > 0, ref.H > 0))
    rws = np.sort(rng.choice(m, size=nsample, replace=False))
    Ar = oracle_block(oracle, data, m, seed, kstar, 0, rows=rws, n=n)
    ref = oracle.nmf(np.asfortranarray(Ar.T), np.asfortranarray(H1.T), np.asfortranarray(W0[rws, :].T), alg,
                     min_iter=1, max_iter=1, normalize=False)
    assert ref.result == 0
    eW = relerr(W1[rws, :], ref.H.T)
    return eH, eW, same


def hals_accurate_iteration_vs_oracle(oracle, A, k, data, seed, kstar, seeds, quant, nsample=48):
    """One accurate-form HALS iteration at full size against the reference's update formulas on sampled rows / columns
    (nmf_solver_hals.hpp:66-117: the W sweep normalises every column inside the sweep, so a sampled row determines its new value
    only up to ONE factor nu_c per column -- the norm over all rows, taken from the device -- and every sampled entry must agree
    on it; nmf_solver_hals.hpp:26-62: the H sweep with the new W is exact on sampled columns).  Returns (largest deviation of a
    sampled W entry from nu_c W1 relative to the column's largest entry, max relative error of the sampled H columns)."""
    from oracle import flatclust as of
    from smallk_amd import NmfSolver, make_options, uniform_host
    m, n = A.height, A.width_global
    with _Env(SMK_NSPLIT="8"):
        s = NmfSolver(A, make_options(m, n, k, "HALS", normalize=False))
    W0, H0 = uniform_host(m, k, seeds[0]), uniform_host(k, n, seeds[1]) * (2.0 / k)
    s.set_factors(W0, H0)
    s.iterate(1)
    assert s.sync() == 0
    W1, H1 = s.factors(normalize=False)
    assert s.product_form()[0] == 8
    s.close()
    assert np.allclose(np.sqrt((W1 * W1).sum(axis=0)), 1.0, rtol=1e-10)
    rng = np.random.default_rng(7)
    rows = np.sort(rng.choice(m, size=nsample, replace=False))
    Ar = oracle_block(oracle, data, m, seed, kstar, quant, rows=rows, n=n)
    G = H0 @ H0.T
    R = Ar @ H0.T
    Wc = W0[rows, :].copy()
    eW = 0.0
    for c in range(k):
        t = Wc[:, c] + (R[:, c] - Wc @ G[:, c]) / G[c, c]
        t[t < 0] = 0.0
        pos = (t > 0) & (W1[rows, c] > 0)
        assert pos.sum() >= 8, c
        nu = np.median(t[pos] / W1[rows, c][pos])
        eW = max(eW, float(np.max(np.abs(t - nu * W1[rows, c])) / np.max(t)))
        Wc[:, c] = W1[rows, c]                       # Gauss-Seidel: later columns see the normalised value
    cols = np.sort(rng.choice(n, size=nsample, replace=False))
    Ac = oracle_block(oracle, data, m, seed, kstar, quant, cols=cols)
    _, _, Hs, _ = of.nnls_hals(Ac, W1, H0[:, cols], 1e-30, 1)     # one sweep, W fixed, no normalisation
    eH = float(np.max(np.abs(H1[:, cols] - Hs)) / max(np.max(np.abs(Hs)), 1e-300))
    return eW, eH
