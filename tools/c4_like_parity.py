"""one-off: BPP k = 64 on 16384 x 8192 fp32 for 20 iterations, fp16 two-term products vs bf16x3 vs the oracle"""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np
import oracle, smallk_amd
smallk_amd.initialize(0)
oracle.set_num_threads(16)
m, n, k, it = 16384, 8192, 64, 20
A = oracle.fill_uniform(m, n, 42, quant=0)
W0 = oracle.fill_uniform(m, k, 43); H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
t0 = time.time(); ref = oracle.nmf(A, W0, H0, "BPP", min_iter=it, max_iter=it); print("oracle %.1f s" % (time.time() - t0), flush=True)
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for ns in ("4", "3"):
    os.environ["SMK_NSPLIT"] = ns
    g = smallk_amd.nmf(A, W0, H0, "BPP", min_iter=it, max_iter=it, storage="f32")
    print("SMK_NSPLIT=%s: result %d  relW %.2e relH %.2e" % (ns, g.result, rel(g.W, ref.W), rel(g.H, ref.H)), flush=True)
