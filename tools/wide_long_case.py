"""One long block-pivoting run against the oracle, error after every few iterations: python tools/wide_long_case.py [k] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, smallk_amd
smallk_amd.initialize(0)
oracle.set_num_threads(16)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
m, n = 1200, 1000
rng = np.random.default_rng(7)
for _ in range(5):      # the sixth draw of tools/wide_long_run.py
    pass
rng = np.random.default_rng(7)
shapes = [(1500, 1200, 160), (1400, 1300, 300), (1500, 1200, 160), (900, 800, 100), (1500, 1200, 200)]
for (mm, nn, kk) in shapes:   # consume the generator exactly as wide_long_run.py does
    r = kk + 2
    (rng.random((mm, r)) * (rng.random((mm, r)) > 0.7)) @ (rng.random((r, nn)) * (rng.random((r, nn)) > 0.7)) + 0.05 * rng.random((mm, nn))
r = k + 2
A = (rng.random((m, r)) * (rng.random((m, r)) > 0.7)) @ (rng.random((r, n)) * (rng.random((r, n)) > 0.7)) + 0.05 * rng.random((m, n))
A = oracle.quantize(A, 0)
W0, H0 = oracle.fill_uniform(m, k, 11), oracle.fill_uniform(k, n, 12)
for it in (1, 2, 4, 8, 12, 16, 20, 25, 30):
    if it > iters: break
    ref = oracle.nmf(A, W0, H0, "BPP", min_iter=it, max_iter=it, tol=1e-14)
    got = smallk_amd.nmf(A, W0, H0, "BPP", min_iter=it, max_iter=it, tol=1e-14)
    ew = np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W); eh = np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H)
    same = np.array_equal(got.H > 0, ref.H > 0)
    print(f"k={k} iterations {it}: relW {ew:.2e} relH {eh:.2e}  supports of H equal: {same}  differing entries {int(((got.H > 0) != (ref.H > 0)).sum())}", flush=True)
