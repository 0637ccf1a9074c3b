"""Time of EACH of the first iterations of a dense NMF run (first-iteration cost against steady state):
   python tools/iter_times.py m n k ALG iters
One line: the iteration times in ms (iterate(1) + sync, host clock), the first, and the median of the second half."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smallk_amd
m, n, k, alg, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
smallk_amd.initialize(0)
rng = np.random.default_rng(0)
A = (rng.random((m, k), dtype=np.float32) @ rng.random((k, n), dtype=np.float32))
D = smallk_amd.DenseMatrix.from_host(A)
s = smallk_amd.NmfSolver(D, smallk_amd.make_options(m, n, k, alg, min_iter=iters, max_iter=iters))
s.set_factors(smallk_amd.uniform_host(m, k, 43), smallk_amd.uniform_host(k, n, 44))
s.iterate(0); s.sync()
ts = []
for i in range(iters):
    t0 = time.perf_counter(); s.iterate(1); rc = s.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0, rc
half = sorted(ts[len(ts) // 2:])
print(f"{m}x{n} k={k} {alg}: first {ts[0]:.2f} ms, second {ts[1]:.2f}, third {ts[2]:.2f}, steady (median of iterations {len(ts)//2 + 1}..{len(ts)}) "
      f"{half[len(half)//2]:.2f} ms; all: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
