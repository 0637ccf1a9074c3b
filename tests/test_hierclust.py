"""CPU tests for HierNMF2: the oracle restatement (oracle/hierclust.py), the host-side pieces of the
product that need no GPU (priority score, option validation), and the reference's own file
writers / SetDiff compiled into oracle/_ref."""
import ctypes as C
import os

import numpy as np
import pytest

from hier_cases import planted

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_hier.so")


def test_priority_product_matches_oracle():
    import smallk_amd
    from oracle import hierclust as oh
    rng = np.random.default_rng(0)
    for t in range(300):
        n = int(rng.integers(2, 80))
        wp = rng.random(n) * (rng.random(n) > 0.3)
        wc = rng.random((n, 2)) * (rng.random((n, 2)) > 0.3)
        if t % 4 == 0:
            wc[:, 0] = np.round(wc[:, 0], 1)          # ties: index order decides (clust_hier_util.hpp:31-45)
            wp = np.round(wp, 1)
        a, b = oh.compute_priority(wp, wc), smallk_amd.hierclust.priority(wp, wc)
        assert a == pytest.approx(b, rel=1e-13, abs=0), (t, a, b)


def test_priority_known_cases():
    from oracle import hierclust as oh
    # fewer than two nonzero parent terms: -3 (clust_hier_util.hpp:128-129)
    assert oh.compute_priority(np.array([0.0, 2.0, 0.0]), np.ones((3, 2))) == -3.0
    # children that rank the terms exactly like the parent score higher than reversed ones
    wp = np.arange(10, 0, -1.0)
    same = np.stack([wp, wp], axis=1)
    rev = np.stack([wp[::-1], wp[::-1]], axis=1)
    assert oh.compute_priority(wp, same) > oh.compute_priority(wp, rev) > 0


def test_setdiff_matches_reference():
    from oracle import hierclust as oh
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref = C.CDLL(REF_SO)
    rng = np.random.default_rng(1)
    for _ in range(50):
        a = np.unique(rng.integers(0, 200, size=60)).astype(np.uint32)
        b = rng.choice(a, size=int(rng.integers(0, len(a))), replace=False)
        b = np.sort(b).astype(np.uint32)
        out = np.zeros(len(a), dtype=np.uint32)
        cnt = ref.ref_setdiff(a.ctypes.data_as(C.POINTER(C.c_uint)), len(a), b.ctypes.data_as(C.POINTER(C.c_uint)),
                              len(b), out.ctypes.data_as(C.POINTER(C.c_uint)))
        assert oh.set_diff(list(a), list(b)) == list(out[:cnt])


def _ref_write(tree, dictionary, fmt, path):
    ref = C.CDLL(REF_SO)
    nodes = tree.nodes
    u = lambda xs: (C.c_uint * len(xs))(*xs)
    i = lambda xs: (C.c_int * len(xs))(*xs)
    offs, terms = [0], []
    for nd in nodes:
        terms += list(nd.term_indices)
        offs.append(len(terms))
    d = (C.c_char_p * len(dictionary))(*[t.encode() for t in dictionary])
    ok = ref.ref_write_tree(path.encode(), 1 if fmt == "JSON" else 0, tree.leaf_doc_count, len(nodes),
                            u([nd.parent for nd in nodes]), i([int(nd.is_left_child) for nd in nodes]),
                            u([nd.left for nd in nodes]), u([nd.right for nd in nodes]),
                            i([len(nd.docs) for nd in nodes]), i(offs), i(terms or [0]), d, len(dictionary))
    assert ok == 1
    return open(path).read()


@pytest.mark.parametrize("fmt", ["JSON", "XML"])
def test_oracle_tree_text_matches_reference_writers(tmp_path, fmt):
    """oracle.tree_text == bytes produced by the reference's hierclust_{json,xml}_writer.cpp."""
    from oracle import hierclust as oh
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    A, _ = planted(60, 90, 3, 5)
    for clusters in (3, 6):                       # 6: the search stops early, unused nodes are written too
        tree, _ = oh.hier_nmf2(A, clusters, seed=11)
        dictionary = [f"term{i}" for i in range(A.shape[0])]
        ref_text = _ref_write(tree, dictionary, fmt, str(tmp_path / f"ref_{clusters}.{fmt.lower()}"))
        assert oh.tree_text(tree, dictionary, fmt) == ref_text


@pytest.mark.parametrize("sparse", [False, True])
def test_oracle_recovers_planted_clusters(sparse):
    from oracle import hierclust as oh
    A, lab = planted(200, 300, 5, 1, sparse=sparse)
    tree, stats = oh.hier_nmf2(A, 5, seed=7)
    asg = np.array(tree.assignments)
    assert len(tree.nodes) == 8 and stats.nmf_count >= 9 and stats.max_count == 0
    assert tree.leaf_doc_count + len(tree.outliers) == 300
    # every planted topic lands in exactly one leaf
    for c in range(5):
        leaves = np.unique(asg[(lab == c) & (asg != oh.NONE)])
        assert len(leaves) == 1
    assert len(np.unique(asg[asg != oh.NONE])) == 5
    # structure invariants of Tree<T> (tree.hpp:214-266)
    for q, nd in enumerate(tree.nodes):
        if nd.left != oh.NONE:
            l, r = tree.nodes[nd.left], tree.nodes[nd.right]
            assert l.parent == q and r.parent == q and l.is_left_child and not r.is_left_child
            assert sorted(l.docs + r.docs) == sorted(nd.docs)
    txt = tree.assignments_text()
    assert txt.splitlines()[0].count(",") == 299 and txt.endswith("\n")


def test_oracle_outlier_branch_and_initializers():
    """A tiny far-off cluster triggers TrialSplit's unbalanced branch; explicit initialisers
    (the --initdir path) are consumed in order and give the same tree twice."""
    from oracle import hierclust as oh
    A, _ = planted(120, 240, 3, 3, tiny=4)
    rng = np.random.default_rng(5)
    inits = [(rng.random((120, 2)), rng.random((2, 240))) for _ in range(40)]
    t1, s1 = oh.hier_nmf2(A, 4, initializers=inits)
    t2, s2 = oh.hier_nmf2(A, 4, initializers=inits)
    assert t1.assignments == t2.assignments and s1.nmf_count == s2.nmf_count
    assert s1.nmf_count > 2 * 3 + 1                # extra ActualSplits from the small-cluster probe
    assert t1.leaf_doc_count + len(t1.outliers) == 240


def test_clust_options_validation():
    import smallk_amd
    from smallk_amd import _lib as L
    from smallk_amd.hierclust import make_clust_options
    v = L.lib().smk_clust_is_valid
    assert v(C.byref(make_clust_options(50, 40, 5)), 1) == 1
    assert v(C.byref(make_clust_options(50, 40, 1)), 1) == 0            # clusters >= 2
    assert v(C.byref(make_clust_options(50, 40, 5, tol=1.0)), 1) == 0
    assert v(C.byref(make_clust_options(50, 40, 5, maxterms=0)), 1) == 0
    assert v(C.byref(make_clust_options(50, 40, 5, unbalanced=1.0)), 1) == 0
    assert v(C.byref(make_clust_options(50, 40, 5, trial_allowance=-1)), 1) == 0
    assert v(C.byref(make_clust_options(0, 40, 5)), 1) == 0
    assert v(C.byref(make_clust_options(0, 40, 5)), 0) == 1             # matrix not validated
    # the product refuses to run without an initialised GPU (no CPU fallback)
    if L.lib().smk_is_initialized() != L.INITIALIZED:
        with pytest.raises(L.SmallkError) as e:
            smallk_amd.hier_nmf2(np.ones((8, 8)), 2)
        assert e.value.code == L.NOTINITIALIZED


@pytest.fixture(scope="module")
def hier_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "hier_golden.npz"))


def _hier_case(name):
    sys_path = os.path.join(ROOT, "tests", "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_hier_golden", os.path.join(sys_path, "make_hier_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name", ["dense5", "dense_outliers", "sparse6", "sparse_outliers"])
def test_oracle_reproduces_committed_hier_fixtures(hier_golden, name):
    """tests/golden/hier_golden.npz (made by tests/golden/make_hier_golden.py) pins the clustering
    oracle: trees, assignments, file texts and the flat factors must come out bit for bit."""
    got = _hier_case(name).run(name)
    for key, val in got.items():
        ref = hier_golden[f"{name}/{key}"]
        if val.dtype.kind in "US":
            assert str(val) == str(ref), key
        elif val.dtype.kind == "f":
            assert np.allclose(val, ref, rtol=1e-12, atol=0), key
        else:
            assert np.array_equal(val, ref), key
