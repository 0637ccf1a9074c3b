"""Generates tests/golden/c5_1m_golden.json: the 8-cluster HierNMF2 tree of the C5-shaped synthetic graph
(1 000 000 nodes, ~16 M stored entries; tests/test_gpu_c5.py:community_graph) as the CPU oracle
(oracle/hierclust.py, big nodes through orc_nmf_sparse) computes it.  About 5 minutes on 8 cores, so it is
run once here and the GPU test compares against the stored summary: structure, per-node document counts and
SHA-256 of the document lists, priorities, top terms, SHA-256 of the assignment vector.

    python tests/golden/make_c5_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint32).tobytes()).hexdigest()


def main():
    from oracle import hierclust as oh
    from test_gpu_c5 import community_graph
    n, deg, clusters, seed = 1_000_000, 16, 8, 1
    A, _ = community_graph(n, deg, 16, 0)
    tree, stats = oh.hier_nmf2(A, clusters, seed=seed, tol=1e-4, max_iter=5000)
    out = {"n": n, "deg": deg, "clusters": clusters, "seed": seed, "nnz": int(A.nnz),
           "nmf_count": stats.nmf_count, "max_count": stats.max_count,
           "assignments_sha256": sha(np.asarray(tree.assignments, dtype=np.uint32)),
           "nodes": [{"parent": int(nd.parent), "left": int(nd.left), "right": int(nd.right), "is_valid": bool(nd.is_valid),
                      "is_left_child": bool(nd.is_left_child), "doc_count": len(nd.docs), "docs_sha256": sha(nd.docs),
                      "priority": float(nd.priority), "term_indices": [int(t) for t in nd.term_indices],
                      "topic_norm": float(np.linalg.norm(nd.topic_vector)), "topic_sum": float(np.sum(nd.topic_vector))}
                     for nd in tree.nodes]}
    with open(os.path.join(ROOT, "tests", "golden", "c5_1m_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("written", stats)


if __name__ == "__main__":
    main()
