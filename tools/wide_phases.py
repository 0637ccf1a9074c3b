"""Phase clocks of the wave-per-column block-pivoting kernel (needs a library built with -DSMK_WIDE_PROFILE):
   python tools/wide_phases.py m n k iters"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smallk_amd
from smallk_amd import _lib as L
m, n, k, iters = (int(x) for x in sys.argv[1:5])
smallk_amd.initialize(0)
rng = np.random.default_rng(0)
A = (rng.random((m, k), dtype=np.float32) @ rng.random((k, n), dtype=np.float32))
W0 = smallk_amd.uniform_host(m, k, 43); H0 = smallk_amd.uniform_host(k, n, 44)
lib = L.lib()
names = ["load + v = Ginv r", "sets / index list", "gather block", "cholesky", "substitutions", "out = base + M u", "scatter / classify", "store", "columns", "exchanges", "sum of block sizes"]
prev = 0
for it in (1, 2, 4, 8, 16):
    if it > iters: break
    r = smallk_amd.nmf(A, W0, H0, "BPP", min_iter=it, max_iter=it)
    out = (ctypes.c_ulonglong * 16)()
    assert lib.smk_debug_wide_profile(out) == 0
    v = list(out)
    tot = sum(v[:8])
    print(f"--- iterations 1..{it}: {r.elapsed_us / 1000:.1f} ms; columns {v[8]}, exchanges per column {v[9] / max(v[8], 1):.2f}, mean block {v[10] / max(v[9], 1):.1f}, exchanges in global scratch {100.0 * v[11] / max(v[9], 1):.1f} %")
    for q in range(8):
        print(f"   {names[q]:22s} {100.0 * v[q] / max(tot, 1):5.1f} %   {v[q] / max(v[9], 1):10.0f} clocks per exchange")
