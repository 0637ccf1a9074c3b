"""NnlsBlockpivot through the C ABI on a fixed family of problems (k = 17 .. 64, solution densities 0 .. 1, warm starts of every
density, 1 .. 4099 columns), results saved to <out.npz>: tests/test_gpu_nnls.py runs it once with SMK_NNLS_G16=0 (a wave per
column, nnls.hip) and once with the default (four columns per wave, nnls_g16.hip) and demands IDENTICAL bits -- the two kernels
perform the same operations in the same order.   python tools/nnls_g16_check.py out.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smallk_amd

smallk_amd.initialize(0)
out = {}
case = 0
for k in (17, 24, 31, 32, 33, 40, 48, 57, 64):
    for fill in (0.0, 0.1, 0.3, 0.5, 0.7, 0.9, 1.0):
        rng = np.random.default_rng(1000 * k + int(fill * 100))
        ncols = int(rng.choice([1, 3, 4, 5, 63, 64, 65, 257, 1000, 4099]))
        m = 4 * k + 5
        W = rng.random((m, k))
        G = np.asfortranarray(W.T @ W)
        B = W.T @ rng.random((m, ncols))
        B -= np.quantile(B, 1.0 - fill) if 0.0 < fill < 1.0 else (np.abs(B).max() * 2 if fill == 0.0 else 0.0)
        X0 = rng.random((k, ncols)) * (rng.random((k, ncols)) < rng.random())
        ok, X, Y = smallk_amd.nnls_blockpivot(G, np.asfortranarray(B), np.asfortranarray(X0))
        out[f"ok{case}"] = np.array([int(ok)])
        out[f"X{case}"] = X
        out[f"Y{case}"] = Y
        case += 1
np.savez(sys.argv[1], **out)
print(f"{case} cases OK")
