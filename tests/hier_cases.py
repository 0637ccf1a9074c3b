"""Synthetic document-term matrices shared by the HierNMF2 tests (CPU and GPU)."""
import numpy as np


def planted(m, n, topics, seed, noise=0.02, sparse=False, tiny=0):
    """Term-document matrix with `topics` planted clusters; `tiny` adds a small far-off cluster
    (exercises the unbalanced / outlier branch of TrialSplit)."""
    rng = np.random.default_rng(seed)
    Wt = np.zeros((m, topics + (1 if tiny else 0)))
    for c in range(Wt.shape[1]):
        idx = rng.choice(m, size=max(3, m // Wt.shape[1]), replace=False)
        Wt[idx, c] = rng.random(len(idx)) + 0.2
    lab = rng.integers(0, topics, size=n)
    if tiny:
        lab[rng.choice(n, size=tiny, replace=False)] = topics
    Ht = np.zeros((Wt.shape[1], n))
    Ht[lab, np.arange(n)] = rng.random(n) + 0.5
    Ht += 0.05 * rng.random(Ht.shape)
    A = Wt @ Ht + noise * rng.random((m, n))
    if sparse:
        import scipy.sparse as sp
        A = A * (A > 0.25)
        return sp.csc_matrix(A), lab
    return np.asfortranarray(A), lab


def tree_arrays(nodes):
    """(parent, left, right, is_left, doc_count, docs, priority, terms) lists from either tree type."""
    out = []
    for nd in nodes:
        out.append(dict(parent=int(nd.parent), left=int(nd.left), right=int(nd.right),
                        is_left=bool(nd.is_left_child), docs=[int(d) for d in nd.docs],
                        priority=float(nd.priority), terms=[int(t) for t in nd.term_indices],
                        valid=bool(nd.is_valid)))
    return out
