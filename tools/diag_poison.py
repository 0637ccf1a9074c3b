import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np
import oracle, smallk_amd, make_golden as mg
smallk_amd.initialize(0)
for (m, n, k, alg, it, st) in [(300, 200, 33, "BPP", 1, "f32"), (96, 64, 5, "BPP", 1, "f32"), (300, 200, 5, "HALS", 1, "bf16"), (700, 400, 65, "BPP", 1, "f32")]:
    A = oracle.fill_uniform(m, n, 42, quant=1 if st == "bf16" else 0)
    W0 = oracle.fill_uniform(m, k, 43); H0 = oracle.fill_uniform(k, n, 44)
    for rep in range(3):
        g = smallk_amd.nmf(A, W0, H0, alg, min_iter=it, max_iter=it, storage=st)
        nanH = np.argwhere(np.isnan(g.H)); nanW = np.argwhere(np.isnan(g.W))
        print(os.environ.get("TAG", ""), m, n, k, alg, "result", g.result, "NaN in H:", len(nanH), "rows", len(set(nanH[:, 0])), "cols", len(set(nanH[:, 1])), "NaN in W:", len(nanW))
