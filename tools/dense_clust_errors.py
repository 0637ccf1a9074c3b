"""Achieved errors of the DENSE clustering rows against the oracle (what the bars in tests/test_gpu_hierclust.py and
tests/test_gpu_flatclust.py are set from): topic vectors (max norm relative to the largest entry), priority scores,
flat factors, NnlsHals.  Run on the GPU box: python3 tools/dense_clust_errors.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
import oracle
import smallk_amd as gpu
from oracle import hierclust as oh, flatclust as of
from hier_cases import planted

gpu.initialize(0)
worst_t = worst_p = 0.0
for storage in ("f32", "bf16"):
    for (m, n, topics, seed, tiny, clusters) in [(200, 300, 5, 1, 0, 5), (120, 240, 3, 3, 4, 4), (96, 150, 4, 9, 0, 7), (161, 30, 4, 5, 0, 8)]:
        A, _ = planted(m, n, topics, seed, tiny=tiny)
        Aq = oracle.quantize(A, 1 if storage == "bf16" else 0)
        res = gpu.hier_nmf2(A, clusters, seed=seed + 100, storage=storage)
        otree, ostats = oh.hier_nmf2(Aq, clusters, seed=seed + 100)
        et = ep = 0.0
        same = len(res.nodes) == len(otree.nodes)
        for q, nd in enumerate(res.nodes):
            if not same or not nd.is_valid or not otree.nodes[q].is_valid:
                continue
            ref = otree.nodes[q].topic_vector
            et = max(et, float(np.max(np.abs(nd.topic_vector - ref)) / max(np.max(np.abs(ref)), 1e-30)))
            pr = otree.nodes[q].priority
            ep = max(ep, abs(nd.priority - pr) / max(abs(pr), 1e-300))
        same = same and list(res.get_assignments()) == list(otree.assignments)
        print(f"hier {storage} {m}x{n} c={clusters}: same_tree={same} topic_rel={et:.2e} prio_rel={ep:.2e} nmf_count={res.nmf_count}/{ostats.nmf_count}")
        worst_t, worst_p = max(worst_t, et), max(worst_p, ep)
print(f"WORST hier dense: topic {worst_t:.2e} priority {worst_p:.2e}")

relmax = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
for k in (3, 6, 20, 150, 300):
    m, n = (130, 210) if k <= 20 else (700, 900)
    rng = np.random.default_rng(k)
    A, _ = planted(m, n, min(k, 6), 17)
    W = np.asfortranarray(rng.random((m, k)) * (rng.random((m, k)) > 0.3))
    H0 = oracle.fill_uniform(k, n, 9)
    rc, Wg, Hg, its = gpu.flatclust.nnls_hals(A, W, H0, tol=1e-6, max_iter=2000)
    ok, Wo, Ho, ito = of.nnls_hals(oracle.quantize(A, 0), W, H0, 1e-6, 2000)
    print(f"nnls_hals k={k}: rc={rc} its={its}/{ito} relW={relmax(Wg, Wo):.2e} relH={relmax(Hg, Ho):.2e}")
for alg in ("HALS", "BPP", "RANK2"):
    k = 2 if alg == "RANK2" else 4
    A, _ = planted(150, 220, k, 13)
    W0 = oracle.fill_uniform(150, k, 1); H0 = oracle.fill_uniform(k, 220, 2)
    res = gpu.flatclust.flatclust(A, W0, H0, alg, maxterms=4, min_iter=5, max_iter=40, tol=1e-9)
    ref = of.flatclust(oracle.quantize(A, 0), W0, H0, alg, min_iter=5, max_iter=40, tol=1e-9)
    print(f"flatclust {alg}: its={res.iteration_count}/{ref.iteration_count} relW={relmax(res.W, ref.W):.2e} relH={relmax(res.H, ref.H):.2e}")
A, _ = planted(120, 200, 4, 21)
res = gpu.hier_nmf2(A, 4, seed=5, flat=True)
otree, _ = oh.hier_nmf2(oracle.quantize(A, 0), 4, seed=5, flat=True)
W, H = res.flat_factors()
print(f"hier+flat: relW={relmax(W, otree.flat_W):.2e} relH={relmax(H, otree.flat_H):.2e} "
      f"fuzzy={float(np.max(np.abs(gpu.flatclust.compute_fuzzy_assignments(H) - of.compute_fuzzy_assignments(otree.flat_H)))):.2e}")
# dense RANK2 by itself, long run, both storages, against the oracle
for storage, q in (("f32", 0), ("bf16", 1)):
    A = oracle.fill_uniform(3000, 1700, 7, quant=q)
    W0 = oracle.fill_uniform(3000, 2, 8); H0 = oracle.fill_uniform(2, 1700, 9)
    ref = oracle.nmf(A, W0, H0, "RANK2", min_iter=200, max_iter=200)
    got = gpu.nmf(A, W0, H0, "RANK2", min_iter=200, max_iter=200, storage=storage)
    fro = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print(f"RANK2 dense {storage} 3000x1700 200 iterations: relW={fro(got.W, ref.W):.2e} relH={fro(got.H, ref.H):.2e}")
