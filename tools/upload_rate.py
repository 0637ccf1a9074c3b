"""Rate of smk_matrix_upload_f64 (host fp64 column-major -> resident matrix incl. the stored transpose) at a given size:
   python3 tools/upload_rate.py m n [storage] [reps]
(profiles/r06_upload_rates.txt also lists the two pipelined variants that were built, measured slower and removed in round 6:
 "mode 1" = pinned staging filled by host threads, "mode 2" = hipHostRegister per chunk; "mode 0" = the loop that stayed)
Prints GB/s of host bytes per repetition; checks the uploaded matrix against the host copy on sampled columns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smallk_amd

m, n = int(sys.argv[1]), int(sys.argv[2])
storage = sys.argv[3] if len(sys.argv) > 3 else "bf16"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
smallk_amd.initialize(0)
G = smallk_amd.DenseMatrix(m, n, storage=storage)
G.fill_uniform(42)
A = G.download()
G.close()
rates = []
for r in range(reps):
    M = smallk_amd.DenseMatrix(m, n, storage=storage)
    t0 = time.perf_counter()
    M.upload(A)
    dt = time.perf_counter() - t0
    rates.append(m * n * 8.0 / dt / 1e9)
    if r == reps - 1:
        B = M.download()
        cols = np.random.default_rng(0).choice(n, size=min(n, 64), replace=False)
        assert np.array_equal(B[:, cols], A[:, cols]), "uploaded matrix differs from the host copy"
        del B
    M.close()
print(f"upload {m}x{n} fp64 ({m * n * 8 / 1e9:.2f} GB) -> {storage}: "
      + " ".join(f"{x:.1f}" for x in rates) + " GB/s", flush=True)
