"""Randomised parity sweep of block principal pivoting above k = 128 (the tile kernels of wide.hip): whole factorisations and
isolated NNLS solves through the C ABI against the oracle.  usage: python tools/fuzz_wide_bpp.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import oracle, smallk_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
smallk_amd.initialize(0)
oracle.set_num_threads(8)
bad = []
t0 = time.time()
for case in range(cases):
    # ---- a whole factorisation ----
    k = int(rng.choice([129, 144, 160, 161, 176, 192, 200, 224, 256, 257, 288, 320, 384, 450, 513]))
    m = int(rng.integers(2 * k, 2 * k + 400)); n = int(rng.integers(2 * k, 2 * k + 400))    # well above k: the Gram matrices stay SPD
    sparse = rng.random() < 0.25
    r = k + 2
    Wt = rng.random((m, r)) * (rng.random((m, r)) > 0.7)           # sparse planted factors: columns of A that differ from each other
    Ht = rng.random((r, n)) * (rng.random((r, n)) > 0.7)           # (a nearly rank-one A gives every column the same support: HH' singular)
    A = Wt @ Ht + 0.05 * rng.random((m, n))
    if sparse:
        A = A * (rng.random((m, n)) < 0.5)
        A[:, A.sum(axis=0) == 0] += 1e-3
    iters = int(rng.integers(1, 5))
    W0 = oracle.fill_uniform(m, k, 100 + case)
    H0 = oracle.fill_uniform(k, n, 200 + case)
    Aq = A if sparse else oracle.quantize(A, 0)
    ref = oracle.nmf(Aq, W0, H0, "BPP", min_iter=iters, max_iter=iters, tol=1e-14)
    if sparse:
        got = smallk_amd.nmf_sparse(sp.csc_matrix(A), W0, H0, "BPP", min_iter=iters, max_iter=iters, tol=1e-14)
    else:
        got = smallk_amd.nmf(A, W0, H0, "BPP", min_iter=iters, max_iter=iters, tol=1e-14)
    desc = f"case {case}: BPP {m}x{n} k={k} {'sparse' if sparse else 'f32'} iters={iters}"
    if got.result != ref.result or got.iteration_count != ref.iteration_count:
        bad.append(desc + f" result {got.result}/{ref.result} iterations {got.iteration_count}/{ref.iteration_count}")
    elif ref.result == 0:
        ew = np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W); eh = np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H)
        if not (ew < 1e-4 and eh < 1e-4): bad.append(desc + f" relW {ew:.2e} relH {eh:.2e}")
    # ---- an isolated solve: random warm start, every passive density, sometimes ill conditioned ----
    k2 = int(rng.integers(129, 700)); ncols = int(rng.integers(1, 24))
    Wm = rng.random((int(rng.choice([k2 + 3, 2 * k2, 4 * k2])), k2))
    G = Wm.T @ Wm
    if rng.random() < 0.3: G = G + np.diag(np.abs(G).sum(axis=1))
    B = Wm.T @ rng.random((Wm.shape[0], ncols))
    fill = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.7, 0.9, 1.0]))
    B -= np.quantile(B, 1.0 - fill) if 0.0 < fill < 1.0 else (np.abs(B).max() * 2 if fill == 0.0 else 0.0)
    X0 = np.asfortranarray(rng.random((k2, ncols)) * (rng.random((k2, ncols)) < rng.random()))
    G = np.asfortranarray(G); B = np.asfortranarray(B)
    oko, Xo, Yo, _ = oracle.nnls_blockpivot(G, B, X0)
    okg, Xg, Yg = smallk_amd.nnls_blockpivot(G, B, X0)
    d2 = f"case {case}: NNLS k={k2} ncols={ncols} fill={fill}"
    if okg != oko: bad.append(d2 + f" ok {okg}/{oko}")
    elif oko:
        sx = max(np.abs(Xo).max(), 1e-300)
        ex = np.abs(Xg - Xo).max() / sx
        if not (ex < 1e-7 and np.array_equal(Xg > 0, Xo > 0)): bad.append(d2 + f" errX {ex:.2e} sets equal {np.array_equal(Xg > 0, Xo > 0)}")
    nfail = nfail + (1 if ref.result != 0 else 0) if case else (1 if ref.result != 0 else 0)
    if (case + 1) % 10 == 0: print(f"{case + 1} cases, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print(f"{cases} cases ({nfail} of the factorisations fail in the oracle too), {len(bad)} bad")
for b in bad[:40]: print("BAD", b)
sys.exit(1 if bad else 0)
