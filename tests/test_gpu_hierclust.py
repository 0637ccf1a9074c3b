"""GPU parity for HierNMF2: the product's tree (smk_clust_dense / smk_clust_sparse through the C ABI)
against the oracle restatement on the same inputs and the same initialiser stream.  The search is
discrete (H(0,c) > H(1,c), priority ordering), so on these well-separated inputs the trees must be
IDENTICAL: structure, per-node documents, top terms, assignments, outliers, factorisation counts.
Topic vectors: these node problems converge slowly (100-1000 RANK2 iterations at tol 1e-4, i.e. a
contraction factor close to 1), which amplifies any product-level difference.  Rounds 1-3 multiplied dense A on
the 16-bit matrix cores (1e-8-class products) and needed 2e-4 / 2e-3 here; since round 4 dense RANK2 takes the
ACCURATE product form (bigprod_f64_k2_kernel: the fp64 product of the stored data), so dense and
sparse alike differ from the oracle by summation order only (measured, tools/dense_clust_errors.py: topic vectors
2e-13, priority scores identical, NnlsHals 1e-14, iteration counts equal).  Bars: topic vectors 1e-9 relative to the
largest entry (dense) / 1e-6 (sparse), priority scores -- a function of the RANKS of the topic vector entries
(clust_hier_util.hpp:105-173) -- 1e-9 both."""
import ctypes as C
import os

import numpy as np
import pytest

from hier_cases import planted, tree_arrays

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_hier.so")


def compare(res, otree, ostats, m, prio_rel=1e-9, topic_rel=1e-9):
    from oracle import hierclust as oh
    a, b = tree_arrays(res.nodes), tree_arrays(otree.nodes)
    assert len(a) == len(b)
    for q, (x, y) in enumerate(zip(a, b)):
        assert x["valid"] == y["valid"], q
        if not y["valid"]:
            continue
        for key in ("parent", "left", "right", "is_left", "docs", "terms"):
            assert x[key] == y[key], (q, key)
        assert x["priority"] == pytest.approx(y["priority"], rel=prio_rel, abs=1e-12), q
    for q, nd in enumerate(res.nodes):
        if nd.is_valid:
            ref = otree.nodes[q].topic_vector
            assert np.max(np.abs(nd.topic_vector - ref)) <= topic_rel * max(np.max(np.abs(ref)), 1e-30), q
    assert list(res.get_assignments()) == list(otree.assignments)
    assert list(res.get_outliers()) == list(otree.outliers)
    assert (res.nmf_count, res.max_count) == (ostats.nmf_count, ostats.max_count)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(200, 300, 5, 1, 0, 5), (120, 240, 3, 3, 4, 4), (96, 150, 4, 9, 0, 7)])
def test_dense_tree_matches_oracle(gpu, storage, case):
    import oracle
    from oracle import hierclust as oh
    m, n, topics, seed, tiny, clusters = case
    A, _ = planted(m, n, topics, seed, tiny=tiny)
    Aq = oracle.quantize(A, 1 if storage == "bf16" else 0)     # what the device holds
    res = gpu.hier_nmf2(A, clusters, seed=seed + 100, storage=storage)
    otree, ostats = oh.hier_nmf2(Aq, clusters, seed=seed + 100)
    compare(res, otree, ostats, m)
    assert res.draws == 2 * ostats.nmf_count                   # W and H per successful attempt


@pytest.mark.parametrize("case", [(300, 400, 6, 2, 0, 6), (150, 260, 3, 4, 5, 5)])
def test_sparse_tree_matches_oracle(gpu, case):
    from oracle import hierclust as oh
    m, n, topics, seed, tiny, clusters = case
    A, _ = planted(m, n, topics, seed, sparse=True, tiny=tiny)
    res = gpu.hier_nmf2(A, clusters, seed=seed)
    otree, ostats = oh.hier_nmf2(A, clusters, seed=seed)
    compare(res, otree, ostats, m, prio_rel=1e-9, topic_rel=1e-6)


def test_initdir_files(gpu, tmp_path):
    """--initdir path: Winit_<i>.csv / Hinit_<i>.csv consumed in order (clust_hier_util.hpp:206-241)."""
    from oracle import hierclust as oh
    from smallk_amd import _lib as L
    m, n = 120, 240
    A, _ = planted(m, n, 3, 3, tiny=4)
    rng = np.random.default_rng(5)
    inits = [(np.asfortranarray(rng.random((m, 2))), np.asfortranarray(rng.random((2, n)))) for _ in range(24)]
    d = str(tmp_path) + "/"
    for i, (W, H) in enumerate(inits, start=1):
        for name, M in (("Winit", W), ("Hinit", H)):
            assert L.lib().smk_write_csv(M.ctypes.data_as(C.POINTER(C.c_double)), M.shape[0], M.shape[0], M.shape[1],
                                         f"{d}{name}_{i}.csv".encode(), 17) == 1
    res = gpu.hier_nmf2(A, 4, initdir=d)
    otree, ostats = oh.hier_nmf2(oracle_quant(A), 4, initializers=inits)
    compare(res, otree, ostats, m)
    # a missing file is a load failure, not a silent random start
    with pytest.raises(L.SmallkError):
        gpu.hier_nmf2(A, 4, initdir=str(tmp_path / "nope") + "/")


def oracle_quant(A):
    import oracle
    return oracle.quantize(A, 0)


@pytest.mark.parametrize("fmt", ["JSON", "XML"])
def test_tree_files(gpu, tmp_path, fmt):
    """smk_tree_write / smk_tree_write_assignments: same bytes as the oracle's text and, where
    oracle/_ref is present, as the reference's own writer objects."""
    from oracle import hierclust as oh
    A, _ = planted(96, 150, 4, 9)
    res = gpu.hier_nmf2(A, 7, seed=3, maxterms=4)
    otree, _ = oh.hier_nmf2(oracle_quant(A), 7, seed=3, maxterms=4)
    dictionary = [f"w{i}" for i in range(96)]
    p = str(tmp_path / f"tree.{fmt.lower()}")
    assert res.write(p, dictionary, fmt)
    text = open(p).read()
    assert text == oh.tree_text(otree, dictionary, fmt)
    pa = str(tmp_path / "assign.csv")
    assert res.write_assignments(pa)
    assert open(pa).read() == otree.assignments_text()
    if os.path.exists(REF_SO):
        from test_hierclust import _ref_write
        assert text == _ref_write(otree, dictionary, fmt, str(tmp_path / "ref.out"))
    # a dictionary shorter than the term count is refused
    assert not res.write(p, dictionary[:3], fmt)


def test_facade_hiernmf2(gpu, tmp_path):
    """smallk::LoadDictionary + HierNmf2 (smallk.cpp:675-862) through the flat handles / SmallkAPI."""
    from oracle import hierclust as oh
    from smallk_amd import SmallkAPI
    A, _ = planted(96, 150, 4, 9)
    dictionary = [f"w{i}" for i in range(96)]
    api = SmallkAPI()
    api.load_matrix(matrix=A)
    api.seed_rng(42)
    # HierNmf2 before a dictionary is loaded: the pysmallk wrapper only prints (smallk_lib.pyx:825-832)
    api.hiernmf2(5)
    api.load_dictionary(dictionary=dictionary)
    from smallk_amd import _lib as L
    assert L.lib().smk_api_set_output_dir(str(tmp_path).encode()) == 0
    api.hiernmf2(5, format="JSON", maxterms=3, tol=1e-4)
    otree, _ = oh.hier_nmf2(oracle_quant(A), 5, seed=42, maxterms=3)
    assert open(tmp_path / "tree_5.json").read() == oh.tree_text(otree, dictionary, "JSON")
    assert open(tmp_path / "assignments_5.csv").read() == otree.assignments_text()
    # dictionary from a file; XML output; unterminated last line is dropped (utils.cpp:220-239)
    dpath = tmp_path / "dict.txt"
    dpath.write_text("\n".join(dictionary) + "\nextra-without-newline")
    api.load_dictionary(filepath=str(dpath))
    api.seed_rng(42)
    api.hiernmf2(5, format="XML", maxterms=3)
    assert open(tmp_path / "tree_5.xml").read() == oh.tree_text(otree, dictionary, "XML")
    # this input stops after 3 splits (4 leaves): the flat step refuses like RunClust does (clust.cpp:53-61)
    assert L.lib().smk_api_hiernmf2_with_flat(5) == 2
    assert b"Insufficient number of leaf nodes" in L.lib().smk_api_last_exception()


def test_gather_cols_dense_and_sparse(gpu):
    """smk_matrix_gather_cols == SubMatrixColsCompact (dense keeps all rows; sparse drops unused rows)."""
    import scipy.sparse as sp
    from smallk_amd import _lib as L
    l = L.lib()
    rng = np.random.default_rng(0)
    m, n = 300, 517
    A = np.asfortranarray(rng.random((m, n)))
    cols = np.sort(rng.choice(n, size=140, replace=False)).astype(np.uint32)
    for storage in (L.STORE_F32, L.STORE_BF16):
        src = gpu.DenseMatrix.from_host(A, storage="bf16" if storage == L.STORE_BF16 else "f32")
        sub = C.c_void_p()
        nh = C.c_int64()
        rows = np.zeros(m, dtype=np.uint32)
        L.check(l.smk_matrix_gather_cols(src._h, cols.ctypes.data_as(C.POINTER(C.c_uint)), len(cols), C.byref(sub),
                                         rows.ctypes.data_as(C.POINTER(C.c_uint)), C.byref(nh)), "gather")
        assert nh.value == m and list(rows) == list(range(m))
        out = np.zeros((m, len(cols)), order="F")
        L.check(l.smk_matrix_download_f64(sub, out.ctypes.data_as(C.POINTER(C.c_double)), m), "download")
        assert np.array_equal(out, src.download()[:, cols])
        l.smk_matrix_destroy(sub)
    # out-of-range column: BAD_PARAM (logic_error in the reference)
    bad = np.array([n], dtype=np.uint32)
    sub = C.c_void_p()
    assert l.smk_matrix_gather_cols(src._h, bad.ctypes.data_as(C.POINTER(C.c_uint)), 1, C.byref(sub), None,
                                    None) == L.BAD_PARAM


def test_priority_device_sort_matches_host(gpu):
    """Above 131072 terms the priority score sorts on the GPU (sort.hip, stable radix sort); the
    permutation -- and therefore the score -- must be the one the host comparator gives, ties, zeros
    and negative zeros included."""
    from oracle import hierclust as oh
    rng = np.random.default_rng(2)
    n = 300_000
    wp = rng.random(n) * (rng.random(n) > 0.3)
    wc = rng.random((n, 2)) * (rng.random((n, 2)) > 0.3)
    wc[:, 0] = np.round(wc[:, 0], 3)                    # many ties
    wc[::7, 1] = -0.0                                   # -0.0 == 0.0 for the reference's comparator
    got = gpu.hierclust.priority(wp, wc)
    assert got == pytest.approx(oh.compute_priority(wp, wc), rel=1e-12)


@pytest.mark.parametrize("m,n,clusters,sparse", [(10, 8, 2, False), (12, 5, 2, False), (30, 9, 3, True), (40, 64, 2, True),
                                                  (7, 40, 6, False)])
def test_small_and_degenerate_inputs(gpu, m, n, clusters, sparse):
    """Tiny matrices: subsets of <= 3 documents are never factored (ActualSplit :399-410), the search
    stops when every leaf priority is negative, two clusters means a single root split."""
    import oracle
    from oracle import hierclust as oh
    A, _ = planted(m, n, 2, 31, sparse=sparse)
    if sparse:
        A = A + 0.01 * __import__("scipy.sparse", fromlist=["eye"]).eye(m, n, format="csc")   # no empty column
    Ad = A if sparse else oracle.quantize(A, 0)
    res = gpu.hier_nmf2(A, clusters, seed=9)
    otree, ostats = oh.hier_nmf2(Ad, clusters, seed=9)
    compare(res, otree, ostats, m, prio_rel=1e-9, topic_rel=(1e-6 if sparse else 1e-9))
    assert len(res.nodes) == 2 * (clusters - 1)


@pytest.mark.parametrize("alg", ["MU", "RANK2"])
def test_sparse_gather_cols_on_device(gpu, alg):
    """Sparse SubMatrixColsCompact assembled on the device (sparse_subset.hip): the kept-row map is the
    reference's, and a factorisation of the gathered matrix -- W'A reads its CSC, AH' the CSC of its
    transpose -- matches the oracle on A[rows][:, cols]."""
    import oracle
    import scipy.sparse as sp
    from smallk_amd import _lib as L, NmfSolver, make_options
    l = L.lib()
    rng = np.random.default_rng(3)
    m, n = 500, 700
    A = sp.random(m, n, density=0.01, random_state=5, format="csc", data_rvs=lambda s: rng.random(s) + 0.1)
    src = gpu.SparseMatrix.from_scipy(A)
    cols = np.sort(rng.choice(n, size=260, replace=False)).astype(np.uint32)
    sub = C.c_void_p()
    nh = C.c_int64()
    rows = np.zeros(m, dtype=np.uint32)
    L.check(l.smk_matrix_gather_cols(src._h, cols.ctypes.data_as(C.POINTER(C.c_uint)), len(cols), C.byref(sub),
                                     rows.ctypes.data_as(C.POINTER(C.c_uint)), C.byref(nh)), "gather")
    Asub = A[:, cols]
    used = np.zeros(m, dtype=bool)
    used[Asub.indices] = True
    want_rows = np.nonzero(used)[0]
    assert nh.value == len(want_rows) and np.array_equal(rows[:nh.value], want_rows)
    D = Asub[want_rows, :].toarray(order="F")
    k = 2 if alg == "RANK2" else 5

    class Handle:            # what NmfSolver needs from a matrix object
        _h, height, ncols = sub, int(nh.value), len(cols)
    W0, H0 = oracle.fill_uniform(Handle.height, k, 1), oracle.fill_uniform(k, Handle.ncols, 2)
    s = NmfSolver(Handle, make_options(Handle.height, Handle.ncols, k, alg, min_iter=4, max_iter=4, tol=1e-12))
    s.set_factors(W0, H0)
    rc, its, _ = s.run()
    W, H = s.factors()
    s.close()
    ref = oracle.nmf(D, W0, H0, alg, min_iter=4, max_iter=4, tol=1e-12)
    assert rc == ref.result == 0 and its == ref.iteration_count
    assert np.max(np.abs(W - ref.W)) <= 1e-9 * np.max(np.abs(ref.W))
    assert np.max(np.abs(H - ref.H)) <= 1e-9 * np.max(np.abs(ref.H))
    l.smk_matrix_destroy(sub)
    # a column list that is not increasing goes through the host cut and gives the same matrix up to
    # the column permutation
    perm = cols[::-1].copy()
    L.check(l.smk_matrix_gather_cols(src._h, perm.ctypes.data_as(C.POINTER(C.c_uint)), len(perm), C.byref(sub),
                                     rows.ctypes.data_as(C.POINTER(C.c_uint)), C.byref(nh)), "gather")
    assert nh.value == len(want_rows) and np.array_equal(rows[:nh.value], want_rows)
    l.smk_matrix_destroy(sub)
    # all-empty selection: "submatrix is the zero matrix" (logic_error in the reference)
    empty = np.nonzero(np.diff(A.indptr) == 0)[0][:3].astype(np.uint32)
    if len(empty):
        assert l.smk_matrix_gather_cols(src._h, empty.ctypes.data_as(C.POINTER(C.c_uint)), len(empty), C.byref(sub), None,
                                        None) == L.BAD_PARAM


@pytest.mark.parametrize("name", ["dense5", "dense_outliers", "sparse6", "sparse_outliers"])
def test_against_committed_fixtures(gpu, tmp_path, name):
    """The product against tests/golden/hier_golden.npz (committed; generator: make_hier_golden.py):
    tree structure, document counts, top terms, assignments, outliers, factorisation counts, the tree
    files byte for byte, and the flat labels where the fixture has a flat step."""
    import importlib.util
    g = np.load(os.path.join(ROOT, "tests", "golden", "hier_golden.npz"))
    spec = importlib.util.spec_from_file_location("make_hier_golden", os.path.join(ROOT, "tests", "golden", "make_hier_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    m, n, topics, dseed, tiny, sparse, clusters, seed, flat = mod.CASES[name]
    A, _ = planted(m, n, topics, dseed, sparse=sparse, tiny=tiny)
    res = gpu.hier_nmf2(A, clusters, seed=seed, maxterms=4, flat=flat)
    nodes = res.nodes
    assert np.array_equal(res.get_assignments(), g[f"{name}/assignments"])
    assert np.array_equal(res.get_outliers(), g[f"{name}/outliers"])
    assert [nd.parent for nd in nodes] == list(g[f"{name}/parent"])
    assert [nd.left for nd in nodes] == list(g[f"{name}/left"]) and [nd.right for nd in nodes] == list(g[f"{name}/right"])
    assert [len(nd.docs) for nd in nodes] == list(g[f"{name}/doc_count"])
    assert (res.nmf_count, res.max_count) == tuple(g[f"{name}/counts"])
    for q, nd in enumerate(nodes):
        if nd.is_valid:
            assert nd.term_indices == list(g[f"{name}/terms"][q]), q
            assert nd.priority == pytest.approx(float(g[f"{name}/priority"][q]), rel=1e-9, abs=1e-12)
    dictionary = [f"w{i}" for i in range(m)]
    for fmt, key in (("JSON", "tree_json"), ("XML", "tree_xml")):
        p = str(tmp_path / f"t.{fmt}")
        assert res.write(p, dictionary, fmt)
        assert open(p).read() == str(g[f"{name}/{key}"])
    pa = str(tmp_path / "a.csv")
    assert res.write_assignments(pa) and open(pa).read() == str(g[f"{name}/assign_text"])
    if flat:
        W, H = res.flat_factors()
        assert np.array_equal(gpu.flatclust.compute_assignments(H), g[f"{name}/flat_labels"])
        assert np.array_equal(gpu.flatclust.top_terms(W, 4), g[f"{name}/flat_terms"])
        # dense: NnlsHals forms W'A once in the accurate product form (round 4), so the fp64 fixture is met like the sparse
        # one in both norms (rounds 1-3: 2e-4 Frobenius / 5e-4 max with the 16-bit product forms; measured now: 2e-15)
        tol = 1e-7
        for X, key in ((W, "flat_W"), (H, "flat_H")):
            G = g[f"{name}/{key}"]
            assert np.linalg.norm(X - G) <= tol * np.linalg.norm(G)
            assert np.max(np.abs(X - G)) <= tol * np.max(np.abs(G))


def test_resident_matrix_is_reused(gpu):
    """smk_clust_resident: the tree search on a matrix that is already in HBM (what smallk::HierNmf2 now
    does: the façade uploads the loaded matrix once and keeps it for every later Nmf / HierNmf2 call)."""
    import oracle
    from oracle import hierclust as oh
    A, _ = planted(150, 260, 3, 4, sparse=True, tiny=5)
    D, _ = planted(120, 200, 4, 21)
    sm = gpu.SparseMatrix.from_scipy(A)
    dm = gpu.DenseMatrix.from_host(D, storage="bf16")
    for seed in (1, 2):                                  # twice on the same resident matrices
        r = gpu.hier_nmf2(sm, 4, seed=seed)
        ot, ost = oh.hier_nmf2(A, 4, seed=seed)
        compare(r, ot, ost, 150, prio_rel=1e-9, topic_rel=1e-6)
        r = gpu.hier_nmf2(dm, 4, seed=seed)
        ot, ost = oh.hier_nmf2(oracle.quantize(D, 1), 4, seed=seed)
        compare(r, ot, ost, 120)
    # and a plain factorisation on the same resident dense matrix afterwards
    W0, H0 = oracle.fill_uniform(120, 6, 1), oracle.fill_uniform(6, 200, 2) / 3
    s = gpu.NmfSolver(dm, gpu.make_options(120, 200, 6, "BPP", min_iter=4, max_iter=4))
    s.set_factors(W0, H0)
    s.run()
    W, H = s.factors()
    ref = oracle.nmf(oracle.quantize(D, 1), W0, H0, "BPP", min_iter=4, max_iter=4)
    assert np.linalg.norm(W - ref.W) / np.linalg.norm(ref.W) < 1e-4


@pytest.mark.parametrize("devices", [2, 4, 8])
@pytest.mark.parametrize("sparse,case", [(True, (300, 400, 6, 2, 0, 6)), (True, (150, 260, 3, 4, 5, 5)), (False, (120, 240, 3, 3, 4, 4)),
                                         (True, (400, 900, 12, 6, 0, 12)), (True, (350, 700, 9, 8, 6, 10))])
def test_two_device_step_gives_the_one_device_tree(gpu, monkeypatch, sparse, case, devices):
    """SMK_CLUST_DEVICES=2 (the two TrialSplits of a step on two devices, clust_hier_generic.hpp:383-517 runs them one after
    the other): the second child is factored by a worker thread with a device context and a copy of A of its own -- on this
    one-GPU box all contexts sit on device 0 (SMK_SHARDS_ON_ONE_GPU=1).  Same tree as the one-device run, node for node and
    bit for bit (it is the same arithmetic on the same initialiser draws), also when outlier trials make the speculation on
    the first child's share of the draws fail (case with tiny clusters).  4 and 8 devices (round 4): the further devices
    factor the children of the NEXT steps speculatively (the best remaining leaves); a speculative step is kept only if the
    sequential search would have made the same split with the same draws, so the tree is still the one-device tree."""
    m, n, topics, seed, tiny, clusters = case
    A, _ = planted(m, n, topics, seed, sparse=sparse, tiny=tiny)
    one = gpu.hier_nmf2(A, clusters, seed=seed + 7)
    monkeypatch.setenv("SMK_CLUST_DEVICES", str(devices))
    monkeypatch.setenv("SMK_SHARDS_ON_ONE_GPU", "1")
    two = gpu.hier_nmf2(A, clusters, seed=seed + 7)
    a, b = tree_arrays(one.nodes), tree_arrays(two.nodes)
    assert len(a) == len(b)
    for q, (x, y) in enumerate(zip(a, b)):
        assert x["valid"] == y["valid"], q
        if not x["valid"]:
            continue
        for key in ("parent", "left", "right", "is_left", "docs", "terms"):
            assert x[key] == y[key], (q, key)
        assert x["priority"] == y["priority"], q
        assert np.array_equal(one.nodes[q].topic_vector, two.nodes[q].topic_vector), q
    assert list(one.get_assignments()) == list(two.get_assignments())
    assert (one.nmf_count, one.max_count, one.draws) == (two.nmf_count, two.max_count, two.draws)
