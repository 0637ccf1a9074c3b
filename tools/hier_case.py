import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, oracle, smallk_amd
from oracle import hierclust as oh
from hier_cases import planted, tree_arrays
smallk_amd.initialize(0)
case = 21
rng = np.random.default_rng(0)
# replay the generator of tools/fuzz_hier.py up to the case
for c in range(case + 1):
    sparse = rng.random() < 0.6
    m, n = int(rng.integers(30, 400)), int(rng.integers(20, 500))
    topics = int(rng.integers(2, 9)); clusters = int(rng.integers(2, 10))
    tiny = int(rng.integers(0, 6)) if n > 60 else 0
    flat = bool(rng.random() < 0.3)
A, _ = planted(m, n, topics, 1000 + case, sparse=sparse, tiny=tiny)
Ad = oracle.quantize(A, 0)
ot, ost = oh.hier_nmf2(Ad, clusters, seed=case)
res = smallk_amd.hier_nmf2(A, clusters, seed=case)
a, b = tree_arrays(res.nodes), tree_arrays(ot.nodes)
print("stats", (res.nmf_count, res.max_count), (ost.nmf_count, ost.max_count))
for q, (x, y) in enumerate(zip(a, b)):
    d = set(x["docs"]) ^ set(y["docs"])
    print(q, x["valid"], y["valid"], x["parent"], y["parent"], len(x["docs"]), len(y["docs"]), "prio %.6f %.6f" % (x["priority"], y["priority"]), "docdiff", sorted(d)[:6], x["terms"] == y["terms"])
