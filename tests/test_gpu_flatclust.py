"""GPU parity for the flat-clustering path: FlatClust / FlatClustSparse, NnlsHals and HierNmf2WithFlat
through the C ABI against the oracle restatement on the same inputs."""
import os

import numpy as np
import pytest

from hier_cases import planted

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("alg", ["HALS", "BPP", "RANK2"])
@pytest.mark.parametrize("sparse", [False, True])
def test_flatclust_matches_oracle(gpu, alg, sparse):
    import oracle
    from oracle import flatclust as of
    k = 2 if alg == "RANK2" else 4
    A, _ = planted(150, 220, k, 13, sparse=sparse)
    W0 = oracle.fill_uniform(150, k, 1)
    H0 = oracle.fill_uniform(k, 220, 2)
    Ad = A if sparse else oracle.quantize(A, 0)
    res = gpu.flatclust.flatclust(A, W0, H0, alg, maxterms=4, min_iter=5, max_iter=40, tol=1e-9)
    ref = of.flatclust(Ad, W0, H0, alg, min_iter=5, max_iter=40, tol=1e-9)
    assert res.result == ref.result == 0 and res.iteration_count == ref.iteration_count
    tol = 1e-8 if sparse else 1e-4
    assert relerr(res.W, ref.W) < tol and relerr(res.H, ref.H) < tol
    # discrete outputs from the same factors
    assert np.array_equal(res.assignments, of.compute_assignments(res.H))
    assert np.array_equal(res.probabilities, of.compute_fuzzy_assignments(res.H))
    assert np.array_equal(res.term_indices, of.top_terms(res.W, 4))
    # and (well separated data) identical to the oracle's
    assert np.array_equal(res.assignments, of.compute_assignments(ref.H))


def test_flatclust_rejects_mu_and_bad_rank2(gpu):
    from smallk_amd import _lib as L
    A = np.ones((16, 16))
    assert gpu.flatclust.flatclust(A, np.ones((16, 3)), np.ones((3, 16)), "MU").result == L.BAD_PARAM
    assert gpu.flatclust.flatclust(A, np.ones((16, 3)), np.ones((3, 16)), "RANK2").result == L.BAD_PARAM


@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize("k", [3, 6, 20, 150, 300, 700])    # the flat step of HierNMF2 runs this with k = number of clusters (<= 1024)
def test_nnls_hals_matches_oracle(gpu, sparse, k):
    import oracle
    from oracle import flatclust as of
    from smallk_amd import _lib as L
    m, n = (130, 210) if k <= 20 else (700, 900)
    rng = np.random.default_rng(k)
    A, _ = planted(m, n, min(k, 6), 17, sparse=sparse)
    W = np.asfortranarray(rng.random((m, k)) * (rng.random((m, k)) > 0.3))
    H0 = oracle.fill_uniform(k, n, 9)
    Ad = A if sparse else oracle.quantize(A, 0)
    rc, Wg, Hg, its = gpu.flatclust.nnls_hals(A, W, H0, tol=1e-6, max_iter=2000)
    ok, Wo, Ho, ito = of.nnls_hals(Ad, W, H0, 1e-6, 2000)
    assert ok and rc == L.OK
    assert its == ito
    tol = 1e-8            # dense too: W'A in the accurate product form (round 4; 2e-4 before, measured now 1e-14)
    assert relerr(Wg, Wo) < tol and relerr(Hg, Ho) < tol
    # iteration limit -> FAILURE, factors left un-normalised (nnls.hpp:311-315)
    rc, Wg, Hg, its = gpu.flatclust.nnls_hals(A, W, H0, tol=1e-14, max_iter=3)
    ok, Wo, Ho, _ = of.nnls_hals(Ad, W, H0, 1e-14, 3)
    assert rc == L.FAILURE and not ok and its == 3
    assert relerr(Wg, Wo) < 1e-12 and relerr(Hg, Ho) < tol


@pytest.mark.parametrize("sparse", [False, True])
def test_hier_with_flat(gpu, sparse):
    from oracle import hierclust as oh, flatclust as of
    import oracle
    A, _ = planted(120, 200, 4, 21, sparse=sparse)
    Ad = A if sparse else oracle.quantize(A, 0)
    res = gpu.hier_nmf2(A, 4, seed=5, flat=True)
    otree, _ = oh.hier_nmf2(Ad, 4, seed=5, flat=True)
    W, H = res.flat_factors()
    tol = 1e-7
    assert relerr(W, otree.flat_W) < tol and relerr(H, otree.flat_H) < tol
    assert list(res.get_assignments()) == list(otree.assignments)
    assert res.draws == otree.draws
    assert np.array_equal(gpu.flatclust.compute_assignments(H), of.compute_assignments(otree.flat_H))
    # fewer leaves than clusters: FLATCLUST_FAILURE, like RunClust (clust.cpp:53-61)
    from smallk_amd import _lib as L
    A5, _ = planted(200, 300, 5, 1)
    with pytest.raises(L.SmallkError) as e:
        gpu.hier_nmf2(A5, 8, seed=7, flat=True)
    assert e.value.code == -6


def test_facade_hiernmf2_with_flat(gpu, tmp_path):
    """smallk::HierNmf2WithFlat (smallk.cpp:865-868): the two tree files plus assignments_flat_N.csv,
    assignments_fuzzy_N.csv, clusters_N.json."""
    import oracle
    from oracle import hierclust as oh, flatclust as of
    from smallk_amd import SmallkAPI, _lib as L
    A, _ = planted(120, 200, 4, 21)
    dictionary = [f"w{i}" for i in range(120)]
    api = SmallkAPI()
    api.load_matrix(matrix=A)
    api.load_dictionary(dictionary=dictionary)
    api.seed_rng(5)
    l = L.lib()
    assert l.smk_api_set_output_dir(str(tmp_path).encode()) == 0
    l.smk_api_set_max_terms(3)
    l.smk_api_set_output_format(1)
    assert l.smk_api_set_hiernmf2_tolerance(1e-4) == 0
    assert l.smk_api_hiernmf2_with_flat(4) == 0, l.smk_api_last_exception()
    otree, _ = oh.hier_nmf2(oracle.quantize(A, 0), 4, seed=5, maxterms=3, flat=True)
    assert open(tmp_path / "tree_4.json").read() == oh.tree_text(otree, dictionary, "JSON")
    assert open(tmp_path / "assignments_4.csv").read() == otree.assignments_text()
    labels = of.compute_assignments(otree.flat_H)
    assert open(tmp_path / "assignments_flat_4.csv").read() == of.assignments_text(labels)
    assert open(tmp_path / "clusters_4.json").read() == of.results_text(labels, of.top_terms(otree.flat_W, 3), dictionary,
                                                                        "JSON", 3, 200, 4)
    got = np.loadtxt(tmp_path / "assignments_fuzzy_4.csv", delimiter=",")
    assert got.shape == (200, 4) and np.allclose(got, of.compute_fuzzy_assignments(otree.flat_H).T, atol=1e-4)     # the file carries 4 significant digits
    # W / H of the flat run are what LockedBufferW/H now expose
    Wf = api.get_W()
    assert Wf.shape == (120, 4) and relerr(Wf, otree.flat_W) < 1e-7


def test_pysmallk_style_classes(gpu, tmp_path):
    """pysmallk's Flatclust / Hierclust classes (smallk_lib.pyx:1080-1420) re-hosted on the C ABI: same
    calls as pysmallk/tests/flatclust.py and hierclust.py, results against the oracle."""
    import oracle
    from oracle import hierclust as oh, flatclust as of
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from pyclust import Flatclust, Hierclust          # examples/pyclust.py: out of the package since round 4
    from test_cli import write_csv
    m, n, k = 120, 200, 4
    A, _ = planted(m, n, k, 21)
    dictionary = [f"w{i}" for i in range(m)]
    (tmp_path / "dict.txt").write_text("\n".join(dictionary) + "\n")
    W0, H0 = oracle.fill_uniform(m, k, 1), oracle.fill_uniform(k, n, 2)
    write_csv(tmp_path / "w0.csv", W0)
    write_csv(tmp_path / "h0.csv", H0)
    write_csv(tmp_path / "a.csv", A)

    f = Flatclust()
    f.load_matrix(filepath=str(tmp_path / "a.csv"))
    f.load_dictionary(filepath=str(tmp_path / "dict.txt"))
    f.cluster(k, infile_W=str(tmp_path / "w0.csv"), infile_H=str(tmp_path / "h0.csv"), algorithm="HALS", maxterms=3,
              verbose=False, min_iter=1, max_iter=50, tol=1e-9)
    ref = of.flatclust(oracle.quantize(A, 0), W0, H0, "HALS", min_iter=1, max_iter=50, tol=1e-9)
    labels = of.compute_assignments(ref.H)
    assert np.array_equal(f.get_assignments(), labels)
    assert np.array_equal(f.get_top_indices(), of.top_terms(ref.W, 3))
    assert f.get_top_terms() == [dictionary[i] for i in of.top_terms(ref.W, 3)]
    out = str(tmp_path) + "/"
    assert f.write_output("assignments", "assignments_fuzzy", "tree", outdir=out, format="JSON")
    assert open(out + "assignments_4.csv").read() == of.assignments_text(labels)
    assert open(out + "tree_4.json").read() == of.results_text(labels, of.top_terms(ref.W, 3), dictionary, "JSON", 3, n, k)

    h = Hierclust()
    h.load_matrix(matrix=A)
    h.load_dictionary(dictionary=dictionary)
    h.cluster(k, maxterms=3, verbose=False, seed=5)
    otree, _ = oh.hier_nmf2(oracle.quantize(A, 0), k, maxterms=3, seed=5)
    assert h.get_assignments() == [(-1 if a == oh.NONE else a) for a in otree.assignments]
    assert h.get_top_indices() == []                                   # only with flat=1 (:1355-1360)
    h.write_output("assign", "htree", "fuzzy", outdir=out, format="XML")
    assert open(out + "htree_4.xml").read() == oh.tree_text(otree, dictionary, "XML")
    assert open(out + "assign_4.csv").read() == otree.assignments_text()
    h.cluster(k, maxterms=3, verbose=False, flat=1, seed=5)
    otree, _ = oh.hier_nmf2(oracle.quantize(A, 0), k, maxterms=3, seed=5, flat=True)
    assert np.array_equal(h.get_assignments(), of.compute_assignments(otree.flat_H))
    assert np.array_equal(h.get_top_indices(), of.top_terms(otree.flat_W, 3))
    # sparse input through the scipy object, as pysmallk/tests/hierclust_inmem.py does with its Sparse class
    As, _ = planted(150, 260, 3, 4, sparse=True)
    hs = Hierclust()
    hs.load_matrix(sparse_matrix=As)
    hs.load_dictionary(dictionary=[f"t{i}" for i in range(150)])
    hs.cluster(5, verbose=False, seed=4)
    ot, _ = oh.hier_nmf2(As, 5, seed=4)
    assert hs.get_assignments() == [(-1 if a == oh.NONE else a) for a in ot.assignments]
