"""Long runs at k <= 64 on data with sparse planted factors (the companion of wide_long_run.py): python tools/long_runs_small_k.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, oracle, smallk_amd
smallk_amd.initialize(0); oracle.set_num_threads(16)
rng = np.random.default_rng(17)
for alg, m, n, k, iters in [("HALS", 1200, 1000, 64, 40), ("HALS", 1200, 1000, 32, 40), ("HALS", 1500, 900, 48, 60), ("MU", 1200, 1000, 64, 60), ("BPP", 1500, 1100, 64, 60), ("BPP", 2000, 1500, 48, 50), ("BPP", 1000, 900, 16, 80)]:
    r = k + 2
    A = (rng.random((m, r)) * (rng.random((m, r)) > 0.7)) @ (rng.random((r, n)) * (rng.random((r, n)) > 0.7)) + 0.05 * rng.random((m, n))
    A = oracle.quantize(A, 0)
    W0, H0 = oracle.fill_uniform(m, k, 21), oracle.fill_uniform(k, n, 22)
    ref = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, tol=1e-14)
    got = smallk_amd.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, tol=1e-14)
    ew = np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W); eh = np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H)
    print(f"{alg} {m}x{n} k={k} {iters} iterations: relW {ew:.2e} relH {eh:.2e}", flush=True)
