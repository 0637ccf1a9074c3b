"""Generates tests/golden/hier_golden.npz: HierNMF2 / flat-clustering fixtures.

The reference ships no tree fixtures in this checkout (its test scripts read the external
smallk_data repository), so these are produced by the oracle restatement (oracle/hierclust.py,
oracle/flatclust.py) on seeded synthetic term-document matrices (tests/hier_cases.py); they pin the
oracle against drift and give the GPU tests a committed target.

    python tests/golden/make_hier_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from hier_cases import planted  # noqa: E402
import oracle  # noqa: E402
from oracle import hierclust as oh, flatclust as of  # noqa: E402

# name: (m, n, topics, data seed, tiny cluster size, sparse, clusters, run seed, flat)
CASES = {
    "dense5": (200, 300, 5, 1, 0, False, 5, 101, False),
    "dense_outliers": (120, 240, 3, 3, 4, False, 4, 103, True),
    "sparse6": (300, 400, 6, 2, 0, True, 6, 2, False),
    "sparse_outliers": (150, 260, 3, 4, 5, True, 3, 4, True),
}


def run(name):
    m, n, topics, dseed, tiny, sparse, clusters, seed, flat = CASES[name]
    A, _ = planted(m, n, topics, dseed, sparse=sparse, tiny=tiny)
    Ad = A if sparse else oracle.quantize(A, 0)             # the device holds fp32
    tree, stats = oh.hier_nmf2(Ad, clusters, seed=seed, maxterms=4, flat=flat)
    dictionary = [f"w{i}" for i in range(m)]
    out = {
        "assignments": np.array(tree.assignments, dtype=np.uint32),
        "outliers": np.array(tree.outliers, dtype=np.uint32),
        "parent": np.array([nd.parent for nd in tree.nodes], dtype=np.uint32),
        "left": np.array([nd.left for nd in tree.nodes], dtype=np.uint32),
        "right": np.array([nd.right for nd in tree.nodes], dtype=np.uint32),
        "doc_count": np.array([len(nd.docs) for nd in tree.nodes], dtype=np.int64),
        "priority": np.array([nd.priority for nd in tree.nodes]),
        "terms": np.array([nd.term_indices if nd.is_valid else [0] * 4 for nd in tree.nodes], dtype=np.int32),
        "counts": np.array([stats.nmf_count, stats.max_count], dtype=np.int64),
        "tree_json": np.array(oh.tree_text(tree, dictionary, "JSON")),
        "tree_xml": np.array(oh.tree_text(tree, dictionary, "XML")),
        "assign_text": np.array(tree.assignments_text()),
    }
    if flat:
        labels = of.compute_assignments(tree.flat_H)
        out["flat_labels"] = labels
        out["flat_terms"] = of.top_terms(tree.flat_W, 4)
        out["flat_W"] = tree.flat_W
        out["flat_H"] = tree.flat_H
    return out


def main():
    blob = {}
    for name in CASES:
        for key, val in run(name).items():
            blob[f"{name}/{key}"] = val
    path = os.path.join(ROOT, "tests", "golden", "hier_golden.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, len(blob), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
