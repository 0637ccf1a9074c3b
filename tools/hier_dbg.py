import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, oracle, smallk_amd
from oracle import hierclust as oh
from hier_cases import planted
smallk_amd.initialize(0)
A, _ = planted(120, 240, 3, 3, tiny=4)
Aq = oracle.quantize(A, 0)
otree, ostats = oh.hier_nmf2(Aq, 4, seed=103)
for q in range(len(otree.nodes)):
    docs = otree.nodes[q].docs
    if len(docs) <= 3: continue
    sub = np.asfortranarray(Aq[:, docs])
    for s in (1, 2):
        W0 = oracle.fill_uniform(120, 2, 1000 + s); H0 = oracle.fill_uniform(2, len(docs), 2000 + s)
        for tol in (1e-4,):
            ro = oracle.nmf(sub, W0, H0, "RANK2", tol=tol, prog_est=0)
            rg = smallk_amd.nmf(sub, W0, H0, "RANK2", tol=tol, prog_est=0)
            dW = np.max(np.abs(ro.W - rg.W)) / np.max(np.abs(ro.W))
            print(q, len(docs), s, "iters", ro.iteration_count, rg.iteration_count, "dW", dW,
                  "metric", ro.metrics[ro.iteration_count - 1] if ro.iteration_count else None)
