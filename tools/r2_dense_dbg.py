import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests/golden")
import numpy as np, oracle, smallk_amd as gpu
import make_golden as mg
gpu.initialize(0)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for (m, n) in [(96, 64), (300, 200), (256, 200), (257, 64), (1000, 700), (64, 300)]:
    for q in (0, 1):
        A = mg.uniform(m, n, 42, q)
        W0 = oracle.fill_uniform(m, 2, 43); H0 = oracle.fill_uniform(2, n, 44)
        ref = oracle.nmf(A, W0, H0, "RANK2", min_iter=1, max_iter=1)
        got = gpu.nmf(A, W0, H0, "RANK2", min_iter=1, max_iter=1, storage="bf16" if q else "f32")
        print(os.environ.get("SMK_NSPLIT", "default"), m, n, q, f"relW {rel(got.W, ref.W):.2e} relH {rel(got.H, ref.H):.2e}", flush=True)
