"""CPU tests for flat clustering: the host post-processing of the product (assignments, fuzzy
assignments, top terms, result files -- no GPU needed) against the oracle restatement and against
the reference's own code compiled into oracle/_ref/libref_flat.so; the oracle's NnlsHals."""
import ctypes as C
import os

import numpy as np
import pytest

from hier_cases import planted

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_flat.so")
_up, _fp, _dp = C.POINTER(C.c_uint), C.POINTER(C.c_float), C.POINTER(C.c_double)


def _H(rng, k, n, ties=False):
    H = np.asfortranarray(rng.random((k, n)))
    if ties:
        H = np.asfortranarray(np.round(H, 1))
        H[:, 0] = 0.5                                  # an all-equal column: first row wins
    return H


@pytest.mark.parametrize("ties", [False, True])
def test_assignments_and_fuzzy(ties):
    import smallk_amd.flatclust as pf
    from oracle import flatclust as of
    rng = np.random.default_rng(3)
    for k, n in ((2, 7), (5, 40), (16, 300)):
        H = _H(rng, k, n, ties) + (0.0 if not ties else 0.05)
        a, p = pf.compute_assignments(H), pf.compute_fuzzy_assignments(H)
        assert np.array_equal(a, of.compute_assignments(H))
        assert np.array_equal(p, of.compute_fuzzy_assignments(H))          # float32, bit for bit
        if os.path.exists(REF_SO):
            ref = C.CDLL(REF_SO)
            ra = np.zeros(n, dtype=np.uint32)
            rp = np.zeros((k, n), dtype=np.float32, order="F")
            ref.ref_compute_assignments(H.ctypes.data_as(_dp), k, k, n, ra.ctypes.data_as(_up))
            ref.ref_compute_fuzzy(H.ctypes.data_as(_dp), k, k, n, rp.ctypes.data_as(_fp))
            assert np.array_equal(a, ra) and np.array_equal(p, rp)
    # k > n is rejected (logic_error in the reference, assignments.hpp:79-80)
    from smallk_amd import _lib as L
    H = _H(rng, 4, 3)
    out = np.zeros(3, dtype=np.uint32)
    assert L.lib().smk_compute_assignments(H.ctypes.data_as(_dp), 4, 4, 3, out.ctypes.data_as(_up)) == L.BAD_PARAM


def test_top_terms():
    import smallk_amd.flatclust as pf
    from oracle import flatclust as of
    rng = np.random.default_rng(4)
    W = np.asfortranarray(rng.random((50, 6)))
    W[:, 2] = np.round(W[:, 2], 1)                       # ties: lower index first
    for mt in (1, 5, 50, 60):                            # 60 > height: only `height` slots are filled
        assert np.array_equal(pf.top_terms(W, mt), of.top_terms(W, mt))
    t = pf.top_terms(W, 3).reshape(6, 3)
    for c in range(6):
        assert list(t[c]) == list(np.argsort(-W[:, c], kind="stable")[:3])
    from smallk_amd import _lib as L
    Wt = np.asfortranarray(rng.random((3, 5)))           # height < width
    out = np.zeros(10, dtype=np.int32)
    assert L.lib().smk_top_terms(2, Wt.ctypes.data_as(_dp), 3, 3, 5, out.ctypes.data_as(C.POINTER(C.c_int))) == L.BAD_PARAM


def _ref_write(tmp, labels, P, terms, dictionary, fmt, maxterms, n, k):
    ref = C.CDLL(REF_SO)
    a = np.ascontiguousarray(labels, dtype=np.uint32)
    p = np.ascontiguousarray(np.asarray(P).T, dtype=np.float32).ravel()
    t = np.ascontiguousarray(terms, dtype=np.int32)
    d = (C.c_char_p * len(dictionary))(*[x.encode() for x in dictionary])
    names = [str(tmp / f"ref_{x}") for x in ("assign", "fuzzy", "result")]
    ref.ref_flat_write_results(names[0].encode(), names[1].encode(), names[2].encode(), a.ctypes.data_as(_up), len(a),
                               p.ctypes.data_as(_fp), len(p), d, len(dictionary), t.ctypes.data_as(C.POINTER(C.c_int)),
                               len(t), 1 if fmt == "JSON" else 0, maxterms, n, k)
    return [open(x).read() for x in names]


@pytest.mark.parametrize("fmt", ["JSON", "XML"])
@pytest.mark.parametrize("empty_cluster", [False, True])
def test_result_files_match_reference_bytes(tmp_path, fmt, empty_cluster):
    """smk_flatclust_write_results == oracle text == the reference's FlatClustWriteResults."""
    import smallk_amd.flatclust as pf
    from oracle import flatclust as of
    rng = np.random.default_rng(7)
    m, n, k, mt = 40, 120, 5, 4
    W = np.asfortranarray(rng.random((m, k)))
    H = _H(rng, k, n)
    if empty_cluster:
        H[3, :] = 0.0                                   # cluster 3 gets no document: doc_count 0, no terms
    res = pf._post(0, W, H, 0, mt)
    dictionary = [f"tok{i}" for i in range(m)]
    files = [str(tmp_path / x) for x in ("a.csv", "f.csv", "r.out")]
    assert res.write_output(*files, dictionary, fmt)
    got = [open(x).read() for x in files]
    labels, P, terms = of.compute_assignments(H), of.compute_fuzzy_assignments(H), of.top_terms(W, mt)
    want = [of.assignments_text(labels), of.fuzzy_text(P), of.results_text(labels, terms, dictionary, fmt, mt, n, k)]
    assert got == want
    if os.path.exists(REF_SO):
        assert got == _ref_write(tmp_path, labels, P, terms, dictionary, fmt, mt, n, k)
    # directory form: the reference's file names
    assert res.write_to_dir(str(tmp_path), dictionary, fmt)
    ext = "json" if fmt == "JSON" else "xml"
    for name in (f"assignments_flat_{k}.csv", f"assignments_fuzzy_{k}.csv", f"clusters_{k}.{ext}"):
        assert (tmp_path / name).exists()
    # too few term indices / a dictionary that is too short are refused
    from smallk_amd import _lib as L
    bad = pf.FlatResult(0, W, H, 0, res.assignments, res.probabilities, res.term_indices[:-1], mt)
    assert not bad.write_output(*files, dictionary, fmt)
    assert not res.write_output(*files, dictionary[:2], fmt)


def test_oracle_nnls_hals_fixed_point():
    """NnlsHals: with W fixed the result satisfies the KKT conditions of min ||A - W H||, H >= 0 to the
    requested tolerance, reproduces a planted H, and reports failure at the iteration limit."""
    from oracle import flatclust as of
    rng = np.random.default_rng(11)
    m, n, k = 80, 150, 6
    W = rng.random((m, k))
    Ht = rng.random((k, n)) * (rng.random((k, n)) > 0.4)
    A = W @ Ht
    ok, Wn, Hn, its = of.nnls_hals(A, W, rng.random((k, n)), 1e-8, 5000)
    assert ok and its > 1
    assert np.allclose(np.sqrt((Wn * Wn).sum(axis=0)), 1.0)             # normalised columns
    assert np.max(np.abs(Wn @ Hn - A)) < 1e-5 * np.max(A)
    grad = (Wn.T @ Wn) @ Hn - Wn.T @ A
    assert np.all(grad[Hn == 0] > -1e-5) and np.max(np.abs(grad[Hn > 0])) < 1e-5
    ok, _, _, its = of.nnls_hals(A, W, rng.random((k, n)), 1e-12, 3)
    assert not ok and its == 3


def test_oracle_hier_flat():
    from oracle import hierclust as oh, flatclust as of
    A, lab = planted(120, 200, 4, 21)
    tree, _ = oh.hier_nmf2(A, 4, seed=5, flat=True)
    assert tree.flat_W.shape == (120, 4) and tree.flat_H.shape == (4, 200)
    labels = of.compute_assignments(tree.flat_H)
    for c in range(4):                                   # flat clusters = planted topics
        assert len(np.unique(labels[lab == c])) == 1
    assert len(np.unique(labels)) == 4
    # the search stops early with fewer leaves than clusters -> FlatclustInitW refuses (tree.hpp:353-359)
    A5, _ = planted(200, 300, 5, 1)
    with pytest.raises(RuntimeError, match="Insufficient"):
        oh.hier_nmf2(A5, 8, seed=7, flat=True)


def test_flatclust_rejects_mu_and_bad_rank2():
    import smallk_amd
    from smallk_amd import _lib as L
    if L.lib().smk_is_initialized() == L.INITIALIZED:
        pytest.skip("GPU present: covered by the gpu test")
    r = smallk_amd.flatclust.flatclust(np.ones((8, 8)), np.ones((8, 2)), np.ones((2, 8)), "HALS")
    assert r.result == L.NOTINITIALIZED and r.assignments is None      # no CPU fallback
