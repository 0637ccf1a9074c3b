"""Long runs above k = 128 against the oracle (the parity cases and sweeps there stop after a few iterations):
   python tools/wide_long_run.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, smallk_amd
smallk_amd.initialize(0)
oracle.set_num_threads(16)
rng = np.random.default_rng(7)
worst = 0.0
for alg, m, n, k, iters in [("BPP", 1500, 1200, 160, 25), ("BPP", 1400, 1300, 300, 20), ("HALS", 1500, 1200, 160, 30), ("HALS", 900, 800, 100, 40),
                            ("MU", 1500, 1200, 200, 40), ("BPP", 1200, 1000, 100, 30)]:
    r = k + 2
    A = (rng.random((m, r)) * (rng.random((m, r)) > 0.7)) @ (rng.random((r, n)) * (rng.random((r, n)) > 0.7)) + 0.05 * rng.random((m, n))
    A = oracle.quantize(A, 0)
    W0, H0 = oracle.fill_uniform(m, k, 11), oracle.fill_uniform(k, n, 12)
    t0 = time.time()
    ref = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, tol=1e-14)
    t1 = time.time()
    got = smallk_amd.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, tol=1e-14)
    ew = np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W); eh = np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H)
    worst = max(worst, ew, eh)
    print(f"{alg} {m}x{n} k={k} {iters} iterations: result {got.result}/{ref.result} relW {ew:.2e} relH {eh:.2e}  (oracle {t1 - t0:.1f} s)", flush=True)
print("worst", f"{worst:.2e}", "PASS" if worst < 1e-4 else "FAIL")
sys.exit(0 if worst < 1e-4 else 1)
