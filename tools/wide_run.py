"""NMF iterations at a rank above 128 on a dense matrix (for timing and rocprofv3 kernel tables):
   python tools/wide_run.py m n k ALG iters [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smallk_amd
m, n, k, alg, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
smallk_amd.initialize(0)
rng = np.random.default_rng(0)
A = (rng.random((m, k), dtype=np.float32) @ rng.random((k, n), dtype=np.float32))
W0 = smallk_amd.uniform_host(m, k, 43); H0 = smallk_amd.uniform_host(k, n, 44)
for rep in range(reps):
    t0 = time.time()
    r = smallk_amd.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters)
    print(f"{m}x{n} k={k} {alg} rep {rep}: result {r.result}, {r.iteration_count} iterations, "
          f"{r.elapsed_us / max(r.iteration_count, 1) / 1000:.3f} ms each (solver clock), wall {time.time() - t0:.2f}s", flush=True)
