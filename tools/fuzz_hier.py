"""Randomised HierNMF2 sweep: sparse inputs (fp64 end to end) must give trees IDENTICAL to the oracle's;
dense inputs are reported (near-ties can legitimately flip).  usage: python tools/fuzz_hier.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
import oracle, smallk_amd
from oracle import hierclust as oh
from hier_cases import planted, tree_arrays

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
smallk_amd.initialize(0)
bad, soft = [], []
t0 = time.time()
for case in range(cases):
    sparse = rng.random() < 0.6
    m, n = int(rng.integers(30, 400)), int(rng.integers(20, 500))
    topics = int(rng.integers(2, 9))
    clusters = int(rng.integers(2, 10))
    tiny = int(rng.integers(0, 6)) if n > 60 else 0
    flat = bool(rng.random() < 0.3)
    A, _ = planted(m, n, topics, 1000 + case, sparse=sparse, tiny=tiny)
    if sparse:
        import scipy.sparse as sp
        A = (A + 1e-3 * sp.eye(m, n, format="csc")).tocsc()
    Ad = A if sparse else oracle.quantize(A, 0)
    desc = f"case {case}: {m}x{n} topics={topics} clusters={clusters} tiny={tiny} {'sparse' if sparse else 'dense'} flat={flat}"
    try:
        ot, ost = oh.hier_nmf2(Ad, clusters, seed=case, flat=flat)
        oerr = None
    except Exception as e:                       # e.g. too few leaves for the flat step
        oerr = str(e)
    try:
        res = smallk_amd.hier_nmf2(A, clusters, seed=case, flat=flat)
        gerr = None
    except smallk_amd._lib.SmallkError as e:
        gerr = str(e)
    if (oerr is None) != (gerr is None):
        (bad if sparse else soft).append(desc + f" oracle error {oerr!r} vs product error {gerr!r}")
        continue
    if oerr is not None:
        continue
    a, b = tree_arrays(res.nodes), tree_arrays(ot.nodes)
    same = all(x["valid"] == y["valid"] and (not y["valid"] or all(x[k] == y[k] for k in ("parent", "left", "right", "docs", "terms")))
               for x, y in zip(a, b)) and list(res.get_assignments()) == list(ot.assignments) \
        and (res.nmf_count, res.max_count) == (ost.nmf_count, ost.max_count)
    if not same:
        (bad if sparse else soft).append(desc + " tree differs")
print(f"{cases} cases in {time.time()-t0:.1f}s; sparse mismatches {len(bad)}, dense mismatches {len(soft)}")
for b in bad + soft:
    print("  ", b)
sys.exit(1 if bad else 0)
