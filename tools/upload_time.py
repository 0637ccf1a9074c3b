import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, smallk_amd
smallk_amd.initialize(0)
for (m, n) in ((8192, 4096), (65536, 4096), (65536, 16384)):
    A = np.asfortranarray(np.random.default_rng(0).random((m, n)))
    for st in ("f32", "bf16"):
        mat = smallk_amd.DenseMatrix(m, n, storage=st)
        mat.upload(A)      # warm
        t0 = time.perf_counter(); mat.upload(A); dt = time.perf_counter() - t0
        print(f"{m}x{n} {st}: upload {dt*1e3:.1f} ms = {A.nbytes/dt/1e9:.1f} GB/s of fp64 host data", flush=True)
        mat.close()
