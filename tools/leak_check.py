import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, torch, smallk_amd, oracle, resource
from hier_cases import planted
smallk_amd.initialize(0)
def free_mb(): return torch.cuda.mem_get_info()[0] / 2**20
def rss_mb(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
A = oracle.fill_uniform(600, 400, 1); W0 = oracle.fill_uniform(600, 8, 2); H0 = oracle.fill_uniform(8, 400, 3) / 4
As, _ = planted(300, 400, 5, 3, sparse=True)
Ad, _ = planted(200, 300, 4, 4)
for rnd in range(8):
    f0, r0 = free_mb(), rss_mb()
    for i in range(150):
        smallk_amd.nmf(A, W0, H0, ("MU", "HALS", "BPP")[i % 3], min_iter=1, max_iter=30, tol=0.01, storage=("f32", "bf16")[i % 2])
    for i in range(15):
        smallk_amd.hier_nmf2(As, 5, seed=i); smallk_amd.hier_nmf2(Ad, 4, seed=i, flat=True)
    torch.cuda.synchronize()
    print(f"round {rnd}: device free {f0:.0f} -> {free_mb():.0f} MiB, host max RSS {r0:.0f} -> {rss_mb():.0f} MiB", flush=True)
