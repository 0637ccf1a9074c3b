"""configs[4] (C5): rank-2 HierNMF2 on a sparse symmetric adjacency matrix, at scale.

* 50 000 nodes: the whole tree against oracle/hierclust.py, node for node (documents, structure, top
  terms, factorisation counts exact; priorities / topic vectors to the sparse-path tolerances of
  tests/test_gpu_hierclust.py).  The oracle factors the big nodes with orc_nmf_sparse (same driver and
  RANK2 solver as the dense restatement, products over the stored entries).
* 1 000 000 nodes, 16 M stored entries: the ROOT factorisation (RANK2 on the full matrix: the two gather
  SpMMs at full length, closed-form solves, per-iteration normalisation, PG-ratio stopping rule) against
  the oracle, iteration count and factors; then the 8-cluster tree node for node against the oracle's tree
  of the same graph (a committed summary: structure, document lists by SHA-256, priorities, top terms) and
  through its invariants (leaves partition the documents, children partition their parent, leaves are made
  of whole planted communities)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def community_graph(n, deg, ncomm, seed):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    comm = rng.integers(0, ncomm, size=n)
    order = np.argsort(comm, kind="stable")
    starts = np.searchsorted(comm[order], np.arange(ncomm + 1))
    half = n * deg // 2
    src = rng.integers(0, n, size=half)
    intra = rng.random(half) < 0.85
    dst = rng.integers(0, n, size=half)
    c = comm[src[intra]]
    dst[intra] = order[starts[c] + (rng.random(intra.sum()) * (starts[c + 1] - starts[c])).astype(np.int64)]
    A = sp.coo_matrix((np.ones(half), (src, dst)), shape=(n, n))
    A = (A + A.T).tocsc()
    A.sum_duplicates()
    A.sort_indices()
    return A, comm


def test_c5_50k_tree_identical_to_oracle(gpu):
    from oracle import hierclust as oh
    A, comm = community_graph(50_000, 16, 16, 0)
    res = gpu.hier_nmf2(A, 8, seed=1, tol=1e-4, max_iter=5000)
    otree, ostats = oh.hier_nmf2(A, 8, seed=1, tol=1e-4, max_iter=5000)
    assert res.nmf_count == ostats.nmf_count and res.max_count == ostats.max_count
    assert len(res.nodes) == len(otree.nodes)
    for q, (g, o) in enumerate(zip(res.nodes, otree.nodes)):
        assert (g.parent, g.left, g.right) == (o.parent, o.left, o.right), q
        assert bool(g.is_valid) == bool(o.is_valid) and bool(g.is_left_child) == bool(o.is_left_child), q
        assert list(g.docs) == list(o.docs), q
        assert list(g.term_indices) == list(o.term_indices), q
        assert g.priority == pytest.approx(o.priority, rel=1e-6, abs=1e-12), q
        assert np.abs(np.asarray(g.topic_vector) - o.topic_vector).max() <= 1e-6 * max(np.abs(o.topic_vector).max(), 1e-300), q
    assert np.array_equal(res.get_assignments(), np.asarray(otree.assignments, dtype=np.uint32))


def test_c5_1m_root_factorisation_and_tree_invariants(gpu):
    import oracle
    from oracle.hierclust import SEED_STRIDE
    n, deg, clusters = 1_000_000, 16, 8
    A, comm = community_graph(n, deg, 16, 0)
    assert A.nnz >= 8_000_000

    # ---- root: RANK2 on the full matrix, same initialisers as the tree search draws (W then H) ----
    seed = 1
    W0 = oracle.fill_uniform(n, 2, seed + SEED_STRIDE * 1)
    H0 = oracle.fill_uniform(2, n, seed + SEED_STRIDE * 2)
    kw = dict(min_iter=5, max_iter=60, tol=1e-4)
    src = gpu.SparseMatrix.from_scipy(A)
    from smallk_amd import NmfSolver, make_options
    s = NmfSolver(src, make_options(n, n, 2, "RANK2", **kw))
    s.set_factors(W0, H0)
    rc, its, _ = s.run()
    Wg, Hg = s.factors()
    s.close()
    ref = oracle.nmf_sparse(A, W0, H0, "RANK2", **kw)
    assert rc == ref.result == 0 and its == ref.iteration_count
    assert np.abs(Wg - ref.W).max() <= 1e-8 * np.abs(ref.W).max()
    assert np.abs(Hg - ref.H).max() <= 1e-8 * np.abs(ref.H).max()

    # ---- the tree: against the oracle's tree of the same graph (tests/golden/c5_1m_golden.json, made once by
    #      tests/golden/make_c5_golden.py -- five minutes of CPU), node for node ----
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c5_1m_golden.json")))
    assert (gold["n"], gold["deg"], gold["clusters"], gold["seed"], gold["nnz"]) == (n, deg, clusters, seed, A.nnz)

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint32).tobytes()).hexdigest()

    res = gpu.hier_nmf2(src, clusters, seed=seed, tol=1e-4, max_iter=5000)
    nodes = res.nodes
    assert (res.nmf_count, res.max_count) == (gold["nmf_count"], gold["max_count"])
    assert len(nodes) == len(gold["nodes"]) == 2 * (clusters - 1)
    for q, (g, o) in enumerate(zip(nodes, gold["nodes"])):
        assert (g.parent, g.left, g.right) == (o["parent"], o["left"], o["right"]), q
        assert bool(g.is_valid) == o["is_valid"] and bool(g.is_left_child) == o["is_left_child"], q
        assert len(g.docs) == o["doc_count"] and sha(g.docs) == o["docs_sha256"], q
        assert list(g.term_indices) == o["term_indices"], q
        assert g.priority == pytest.approx(o["priority"], rel=1e-6, abs=1e-12), q
        assert float(np.linalg.norm(g.topic_vector)) == pytest.approx(o["topic_norm"], rel=1e-6), q
        assert float(np.sum(g.topic_vector)) == pytest.approx(o["topic_sum"], rel=1e-6), q
    assert sha(res.get_assignments()) == gold["assignments_sha256"]

    # ---- and through its invariants ----
    NONE = 0xFFFFFFFF
    leaves = [q for q, nd in enumerate(nodes) if nd.is_valid and nd.left == NONE]
    assert len(leaves) == clusters
    seen = np.zeros(n, dtype=np.int32)
    for q in leaves:
        seen[np.asarray(nodes[q].docs, dtype=np.int64)] += 1
    outliers = np.asarray(res.get_outliers(), dtype=np.int64)
    seen[outliers] += 1
    assert (seen == 1).all()                               # leaves (+ outliers) partition the documents
    for q, nd in enumerate(nodes):
        if nd.is_valid and nd.left != NONE:
            l, r = nodes[nd.left], nodes[nd.right]
            assert l.parent == q and r.parent == q and l.is_left_child and not r.is_left_child
            kids = np.sort(np.concatenate([np.asarray(l.docs), np.asarray(r.docs)]))
            mine = np.sort(np.asarray(nd.docs))
            assert len(kids) <= len(mine) and np.isin(kids, mine).all()     # children partition (a subset of) the parent
            assert len(np.unique(kids)) == len(kids)
    # 16 planted communities in 8 leaves: every leaf is made of whole communities (two each when balanced)
    asg = res.get_assignments()
    for leaf in np.unique(asg[asg != NONE]):
        share = np.sort(np.bincount(comm[asg == leaf], minlength=16))[::-1] / float((asg == leaf).sum())
        assert share[:3].sum() > 0.9
    src.close()
