import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, smallk_amd
smallk_amd.initialize(0)
for (m, n, k, alg, st) in ((8192, 4096, 16, "BPP", "f32"), (65536, 16384, 32, "HALS", "bf16"), (2048, 1024, 8, "MU", "f32"), (512, 256, 8, "HALS", "f32")):
    A = smallk_amd.DenseMatrix(m, n, storage=st); A.fill_uniform(1)
    W0 = smallk_amd.uniform_host(m, k, 2); H0 = smallk_amd.uniform_host(k, n, 3) * (2.0 / k)
    iters = 200
    s = smallk_amd.NmfSolver(A, smallk_amd.make_options(m, n, k, alg, min_iter=iters, max_iter=iters)); s.set_factors(W0, H0)
    s.iterate(5); s.sync()
    t0 = time.perf_counter(); s.iterate(iters); s.sync(); t_free = (time.perf_counter() - t0) / iters
    s.close()
    s = smallk_amd.NmfSolver(A, smallk_amd.make_options(m, n, k, alg, min_iter=1, max_iter=iters, tol=1e-300)); s.set_factors(W0, H0)
    t0 = time.perf_counter(); rc, it, us = s.run(); t_chk = (time.perf_counter() - t0) / max(it, 1)
    s.close(); A.close()
    print(f"{m}x{n} k={k} {alg}: {t_free*1e6:.0f} us/iter unchecked, {t_chk*1e6:.0f} us/iter with the stopping rule every iteration ({it} iters)", flush=True)
