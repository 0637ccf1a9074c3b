"""diagnostic: bitwise repeatability of short sparse / dense runs while a second process holds a GPU context"""
import sys, os, subprocess, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np
holder = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0,'.'); import smallk_amd, time; smallk_amd.initialize(0); time.sleep(600)"])
time.sleep(8)
import oracle, smallk_amd
from smallk_amd import solver as S
from hier_cases import planted
smallk_amd.initialize(0)
A, _ = planted(400, 600, 40, 77, sparse=True)
reps = int(sys.argv[1])
try:
    for (alg, k, sparse) in [("BPP", 32, True), ("BPP", 8, True), ("HALS", 16, True), ("MU", 16, True), ("BPP", 32, False), ("HALS", 32, False)]:
        W0 = oracle.fill_uniform(400, k, 5); H0 = oracle.fill_uniform(k, 600, 6)
        first = None; bad = 0; fails = 0
        for r in range(reps):
            try:
                g = S.nmf_sparse(A, W0, H0, alg, min_iter=3, max_iter=3) if sparse else smallk_amd.nmf(np.asfortranarray(A.toarray()), W0, H0, alg, min_iter=3, max_iter=3)
            except Exception as e:
                fails += 1; continue
            if g.result != 0: fails += 1; continue
            if first is None: first = (g.W.copy(), g.H.copy())
            elif not (np.array_equal(first[0], g.W) and np.array_equal(first[1], g.H)):
                bad += 1
                if bad <= 3: print("  mismatch rep", r, "W", np.abs(first[0]-g.W).max(), "H", np.abs(first[1]-g.H).max(), flush=True)
        print(alg, k, "sparse" if sparse else "dense", ": mismatching runs", bad, "failed runs", fails, "of", reps, flush=True)
finally:
    holder.terminate()
