"""time per iteration of the general-rank path (k > 128) on a mid-size dense problem"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import smallk_amd
from smallk_amd import DenseMatrix, NmfSolver, make_options
smallk_amd.initialize(0)
m, n = 16384, 8192
for k in (128, 192, 256, 512):
    for alg in ("MU", "HALS", "BPP"):
        D = DenseMatrix(m, n, storage="f32"); D.fill_uniform(42)
        W0 = smallk_amd.uniform_host(m, k, 43); H0 = smallk_amd.uniform_host(k, n, 44) * (2.0 / k)
        it = 4
        s = NmfSolver(D, make_options(m, n, k, alg, min_iter=it + 1, max_iter=it + 1))
        s.set_factors(W0, H0)
        s.iterate(1); s.sync()
        t0 = time.perf_counter(); s.iterate(it); rc = s.sync(); dt = (time.perf_counter() - t0) / it
        print(f"{m}x{n} k={k} {alg}: {dt*1e3:.2f} ms / iteration (rc {rc})", flush=True)
        s.close(); D.close()
