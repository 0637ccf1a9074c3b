"""The product's host-side bookkeeping against the reference's OWN code, compiled in place from
/root/reference into oracle/_ref/libref_sparse.so (`make -C oracle ref`; the .so travels to the GPU box,
the sources never enter this repository):

  smk_is_valid                  <-> IsValid                 common/src/nmf_options.cpp:23-112
  smk_csc_transpose             <-> Transpose(SparseMatrix) common/include/sparse_matrix_ops.hpp:36-127
  smk_csc_subset_cols_compact   <-> SubMatrixColsCompact    common/include/sparse_matrix_impl.hpp:478-592
  smk_load_matrix_market        <-> LoadMatrixMarketFile    common/include/sparse_matrix_io.hpp:117-259
  oracle.hierclust._Source      <-> SubMatrixColsCompact    (the oracle's restatement is pinned too)

Everything here is integer / copy work: the bar is exact equality.
"""
import ctypes as C
import itertools
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_sparse.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libref_sparse.so not built (needs /root/reference)")

UP = C.POINTER(C.c_uint)
DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def ref():
    r = C.CDLL(REF)
    r.ref_is_valid.restype = C.c_int
    r.ref_is_valid.argtypes = [C.c_double] + [C.c_int] * 12
    for name in ("ref_sm_from_triplets", "ref_sm_from_csc", "ref_sm_transpose", "ref_sm_submatrix_cols_compact",
                 "ref_sm_load_matrix_market"):
        getattr(r, name).restype = C.c_void_p
    r.ref_sm_from_triplets.argtypes = [C.c_uint, C.c_uint, C.c_uint, UP, UP, DP]
    r.ref_sm_from_csc.argtypes = [C.c_uint, C.c_uint, C.c_uint, UP, UP, DP]
    r.ref_sm_transpose.argtypes = [C.c_void_p]
    r.ref_sm_submatrix_cols_compact.argtypes = [C.c_void_p, UP, C.c_uint, UP, UP, UP]
    r.ref_sm_load_matrix_market.argtypes = [C.c_char_p, UP, UP, UP]
    r.ref_sm_free.argtypes = [C.c_void_p]
    for name in ("ref_sm_height", "ref_sm_width", "ref_sm_size"):
        getattr(r, name).restype = C.c_uint
        getattr(r, name).argtypes = [C.c_void_p]
    r.ref_sm_copy_out.argtypes = [C.c_void_p, UP, UP, DP]
    r.ref_is_sparse_file.argtypes = [C.c_char_p]
    r.ref_is_dense_file.argtypes = [C.c_char_p]
    return r


@pytest.fixture(scope="module")
def lib():
    from smallk_amd import _lib as L
    return L.lib()


def up(a):
    return a.ctypes.data_as(UP)


def dp(a):
    return a.ctypes.data_as(DP)


def ref_csc(ref, h):
    """(height, width, colptr, rowidx, data) of a reference SparseMatrix handle."""
    hh, ww, nz = ref.ref_sm_height(h), ref.ref_sm_width(h), ref.ref_sm_size(h)
    cp = np.zeros(ww + 1, dtype=np.uint32)
    ri = np.zeros(max(nz, 1), dtype=np.uint32)
    va = np.zeros(max(nz, 1))
    ref.ref_sm_copy_out(h, up(cp), up(ri), dp(va))
    return hh, ww, cp, ri[:nz], va[:nz]


def random_csc(rng, m, n, density, empty_cols=True):
    import scipy.sparse as sp
    A = sp.random(m, n, density=density, random_state=int(rng.integers(1 << 30)), format="csc",
                  data_rvs=lambda s: rng.random(s) + 0.05)
    A.sort_indices()
    return A.indptr.astype(np.uint32), A.indices.astype(np.uint32), A.data.astype(np.float64)


# ------------------------------------------------------------------------------------------------
def test_is_valid_matches_reference(ref, lib):
    from smallk_amd import _lib as L
    tols = [-1.0, 0.0, 1e-300, 0.005, 0.999999, 1.0, 2.0, float("nan")]
    algs = [-1, 0, 1, 2, 3, 4]
    progs = [-1, 0, 1, 2]
    shapes = [(0, 5), (5, 0), (-3, 4), (7, 9), (9, 7), (1, 1)]
    ks = [-1, 0, 1, 2, 7, 9, 10, 65, 200]
    its = [(-1, 5), (0, 5), (5, 0), (5, -2), (1, 1), (10, 5)]
    tcs = [-1, 0, 1, 3]
    rng = np.random.default_rng(11)
    cases = list(itertools.product(tols[:4] + tols[5:6], algs, progs[1:3], shapes[3:4], ks, its[4:5], tcs[2:3]))
    cases += list(itertools.product(tols, algs[1:2], progs, shapes, ks[2:5], its, tcs))
    for _ in range(3000):
        cases.append((tols[rng.integers(len(tols))], algs[rng.integers(len(algs))], progs[rng.integers(len(progs))],
                      shapes[rng.integers(len(shapes))], ks[rng.integers(len(ks))], its[rng.integers(len(its))],
                      tcs[rng.integers(len(tcs))]))
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)           # both sides print the reference's messages on stderr
    try:
        bad = []
        for tol, alg, prog, (h, w), k, (mn, mx), tc in cases:
            for vm in (0, 1):
                o = L.Options(tol, alg, prog, h, w, k, mn, mx, tc, 0, 0, 1)
                got = lib.smk_is_valid(C.byref(o), vm)
                want = ref.ref_is_valid(tol, alg, prog, h, w, k, mn, mx, tc, 0, 0, 1, vm)
                if got != want:
                    bad.append((tol, alg, prog, h, w, k, mn, mx, tc, vm, got, want))
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
        os.close(saved)
    assert not bad, bad[:5]
    assert len(cases) > 5000


def test_is_valid_messages_match_reference(ref, lib, capfd):
    """Same diagnostic text on stderr (nmf_options.cpp prints through cerr)."""
    from smallk_amd import _lib as L
    for args in [(0.005, 3, 0, 5, 5, 0, 1, 1, 1), (0.005, 3, 0, 0, 5, 2, 1, 1, 1), (0.005, 3, 0, 5, 0, 2, 1, 1, 1),
                 (0.005, 3, 0, 5, 5, 6, 1, 1, 1), (1.5, 3, 0, 5, 5, 2, 1, 1, 1), (0.005, 3, 0, 5, 5, 2, 0, 1, 1),
                 (0.005, 3, 0, 5, 5, 2, 1, 0, 1), (0.005, 3, 0, 5, 5, 2, 1, 1, 0), (0.005, 9, 0, 5, 5, 2, 1, 1, 1),
                 (0.005, 2, 0, 5, 5, 3, 1, 1, 1), (0.005, 3, 7, 5, 5, 2, 1, 1, 1)]:
        tol, alg, prog, h, w, k, mn, mx, tc = args
        capfd.readouterr()
        assert ref.ref_is_valid(tol, alg, prog, h, w, k, mn, mx, tc, 0, 0, 1, 1) == 0
        want = capfd.readouterr().err
        o = L.Options(tol, alg, prog, h, w, k, mn, mx, tc, 0, 0, 1)
        assert lib.smk_is_valid(C.byref(o), 1) == 0
        got = capfd.readouterr().err
        assert got.strip() == want.strip() and want.strip()


@pytest.mark.parametrize("m,n,density", [(1, 1, 1.0), (5, 1, 0.6), (1, 7, 0.6), (40, 30, 0.1), (300, 517, 0.02), (64, 64, 0.0),
                                          (2000, 1500, 0.004)])
def test_transpose_matches_reference(ref, lib, m, n, density):
    rng = np.random.default_rng(m * 1000 + n)
    cp, ri, va = random_csc(rng, m, n, density)
    nz = len(ri)
    h = ref.ref_sm_from_csc(m, n, nz, up(cp), up(ri if nz else np.zeros(1, np.uint32)), dp(va if nz else np.zeros(1)))
    t = ref.ref_sm_transpose(h)
    th, tw, tcp, tri, tva = ref_csc(ref, t)
    assert (th, tw) == (n, m)
    ocp = np.zeros(m + 1, dtype=np.uint32)
    ori = np.zeros(max(nz, 1), dtype=np.uint32)
    ova = np.zeros(max(nz, 1))
    assert lib.smk_csc_transpose(m, n, up(cp), up(ri if nz else ori), dp(va if nz else ova), up(ocp), up(ori), dp(ova)) == 0
    assert np.array_equal(ocp, tcp) and np.array_equal(ori[:nz], tri) and np.array_equal(ova[:nz], tva)
    ref.ref_sm_free(t)
    ref.ref_sm_free(h)


def product_subset(lib, m, n, cp, ri, va, cols):
    nh, nz = C.c_int64(), C.c_int64()
    rc = lib.smk_csc_subset_cols_compact(m, n, up(cp), up(ri), dp(va), up(cols), len(cols), None, None, None, None, None,
                                         C.byref(nh), C.byref(nz))
    if rc:
        return rc, None
    ocp = np.zeros(len(cols) + 1, dtype=np.uint32)
    ori = np.zeros(nz.value, dtype=np.uint32)
    ova = np.zeros(nz.value)
    o2n = np.zeros(m, dtype=np.uint32)
    n2o = np.zeros(m, dtype=np.uint32)
    rc = lib.smk_csc_subset_cols_compact(m, n, up(cp), up(ri), dp(va), up(cols), len(cols), up(ocp), up(ori), dp(ova),
                                         up(o2n), up(n2o), C.byref(nh), C.byref(nz))
    return rc, (nh.value, ocp, ori, ova, o2n, n2o[:nh.value])


@pytest.mark.parametrize("m,n,density,pick", [(30, 20, 0.2, 7), (300, 517, 0.02, 140), (1000, 800, 0.003, 500), (50, 40, 0.5, 40),
                                               (200, 100, 0.01, 1)])
def test_submatrix_cols_compact_matches_reference(ref, lib, m, n, density, pick):
    from oracle import hierclust as oh
    import scipy.sparse as sp
    rng = np.random.default_rng(pick * 7 + m)
    cp, ri, va = random_csc(rng, m, n, density)
    nz = len(ri)
    h = ref.ref_sm_from_csc(m, n, nz, up(cp), up(ri), dp(va))
    nonempty = np.nonzero(np.diff(cp.astype(np.int64)) > 0)[0]
    for trial in range(6):
        cols = rng.choice(n, size=pick, replace=False).astype(np.uint32)
        if trial % 2 == 0:
            cols = np.sort(cols)                 # HierNMF2's document lists are increasing
        if not np.isin(cols, nonempty).any():
            cols[0] = nonempty[0]
        o2n = np.zeros(m, dtype=np.uint32)
        n2o = np.zeros(m, dtype=np.uint32)
        nh = C.c_uint()
        sub = ref.ref_sm_submatrix_cols_compact(h, up(cols), len(cols), up(o2n), up(n2o), C.byref(nh))
        assert sub
        sh, sw, scp, sri, sva = ref_csc(ref, sub)
        rc, got = product_subset(lib, m, n, cp, ri, va, cols)
        assert rc == 0
        gh, gcp, gri, gva, go2n, gn2o = got
        assert (gh, len(cols)) == (sh, sw) == (nh.value, len(cols))
        assert np.array_equal(gcp, scp) and np.array_equal(gri, sri) and np.array_equal(gva, sva)
        assert np.array_equal(gn2o, n2o[:nh.value])
        kept = go2n != 0xFFFFFFFF
        assert np.array_equal(go2n[kept], o2n[kept]) and kept.sum() == nh.value
        # the oracle's restatement (used for every HierNMF2 parity test) is the same matrix
        A = sp.csc_matrix((va, ri, cp), shape=(m, n))
        D, rows = oh._Source(A).subset(cols)
        assert np.array_equal(rows, gn2o)
        assert np.array_equal(D, sp.csc_matrix((sva, sri, scp), shape=(sh, sw)).toarray())
        ref.ref_sm_free(sub)
    ref.ref_sm_free(h)


def test_submatrix_of_empty_columns_is_an_error_on_both_sides(ref, lib):
    m, n = 12, 6
    cp = np.array([0, 2, 2, 2, 3, 3, 3], dtype=np.uint32)
    ri = np.array([1, 5, 7], dtype=np.uint32)
    va = np.array([1.0, 2.0, 3.0])
    h = ref.ref_sm_from_csc(m, n, 3, up(cp), up(ri), dp(va))
    cols = np.array([1, 2, 4], dtype=np.uint32)
    nh = C.c_uint()
    assert not ref.ref_sm_submatrix_cols_compact(h, up(cols), 3, None, None, C.byref(nh))     # throws logic_error
    rc, _ = product_subset(lib, m, n, cp, ri, va, cols)
    assert rc == -3
    ref.ref_sm_free(h)


MM_FILES = {
    "general_real": "%%MatrixMarket matrix coordinate real general\n% a comment\n4 5 6\n1 1 1.5\n4 5 -2.25\n2 3 1e-3\n3 3 7\n1 5 0.125\n2 1 3\n",
    "unsorted_dupes": "%%MatrixMarket matrix coordinate real general\n3 3 5\n3 3 1.0\n1 1 2.0\n3 3 4.0\n2 1 0.5\n1 1 0.25\n",
    "explicit_zero": "%%MatrixMarket matrix coordinate real general\n3 2 3\n1 1 0.0\n2 2 5.0\n3 1 1.0\n",
    "integer": "%%MatrixMarket matrix coordinate integer general\n3 4 4\n1 2 3\n3 4 -7\n2 2 1\n3 1 12\n",
    "pattern": "%%MatrixMarket matrix coordinate pattern general\n4 4 5\n1 1\n2 3\n4 4\n3 1\n1 4\n",
    "symmetric": "%%MatrixMarket matrix coordinate real symmetric\n4 4 5\n1 1 2.0\n2 1 3.0\n4 2 -1.0\n3 3 4.0\n4 4 0.5\n",
    "skew": "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 1.5\n3 1 -2.0\n",
    "pattern_symmetric": "%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n2 1\n3 3\n3 2\n",
    "blank_and_comments": "%%MatrixMarket matrix coordinate real general\n%c1\n%c2\n\n2 2 2\n1 2 9.0\n2 1 8.0\n",
    "empty_columns": "%%MatrixMarket matrix coordinate real general\n5 6 2\n2 2 1.0\n5 5 2.0\n",
    "dense_array": "%%MatrixMarket matrix array real general\n2 2\n1.0\n2.0\n3.0\n4.0\n",
    "complex": "%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1.0 2.0\n",
    "bad_banner": "%MatrixMarket matrix coordinate real general\n2 2 1\n1 1 1.0\n",
    "truncated": "%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1 1.0\n2 2 2.0\n",
    "hermitian": "%%MatrixMarket matrix coordinate real hermitian\n2 2 1\n1 1 1.0\n",
}


@pytest.mark.parametrize("name", sorted(MM_FILES))
def test_matrix_market_loader_matches_reference(ref, lib, tmp_path, name):
    path = tmp_path / (name + ".mtx")
    path.write_text(MM_FILES[name])
    p = str(path).encode()
    h_, w_, nz_ = C.c_uint(), C.c_uint(), C.c_uint()
    handle = ref.ref_sm_load_matrix_market(p, C.byref(h_), C.byref(w_), C.byref(nz_))
    gh, gw, gnz = C.c_uint(), C.c_uint(), C.c_uint()
    ok = lib.smk_load_matrix_market(p, C.byref(gh), C.byref(gw), C.byref(gnz), None, None, None)
    assert bool(ok) == bool(handle), name
    assert ref.ref_is_sparse_file(p) == 1 and ref.ref_is_dense_file(p) == 0
    if not handle:
        return
    rh, rw, rcp, rri, rva = ref_csc(ref, handle)
    # (the reference's `nnz` out-parameter repeats the header count; the product reports the stored entries,
    #  which is what sizes the arrays: SparseMatrix::Size() after mirroring symmetric / skew files)
    assert (gh.value, gw.value, gnz.value) == (rh, rw, len(rri)) and (h_.value, w_.value) == (rh, rw)
    cp = np.zeros(rw + 1, dtype=np.uint32)
    ri = np.zeros(max(len(rri), 1), dtype=np.uint32)
    va = np.zeros(max(len(rri), 1))
    assert lib.smk_load_matrix_market(p, C.byref(gh), C.byref(gw), C.byref(gnz), up(cp), up(ri), dp(va)) == 1
    assert np.array_equal(cp, rcp) and np.array_equal(ri[:len(rri)], rri) and np.array_equal(va[:len(rri)], rva)
    ref.ref_sm_free(handle)


def test_matrix_market_random_files_match_reference(ref, lib, tmp_path):
    """Larger random coordinate files, entries in random order, every symmetry class."""
    rng = np.random.default_rng(5)
    for case, sym in enumerate(["general", "symmetric", "skew-symmetric", "general"]):
        m = int(rng.integers(20, 200))
        n = m if sym != "general" else int(rng.integers(20, 200))
        cnt = int(rng.integers(50, 600))
        r = rng.integers(1, m + 1, cnt)
        c = rng.integers(1, n + 1, cnt)
        if sym != "general":
            r, c = np.maximum(r, c), np.minimum(r, c)
            if sym == "skew-symmetric":
                keep = r != c
                r, c = r[keep], c[keep]
        pairs = sorted(set(zip(r.tolist(), c.tolist())))         # MatrixMarket files do not repeat coordinates
        rng.shuffle(pairs)
        lines = ["%%MatrixMarket matrix coordinate real " + sym, f"{m} {n} {len(pairs)}"]
        lines += [f"{a} {b} {rng.random() + 0.01:.17g}" for a, b in pairs]
        path = tmp_path / f"rand{case}.mtx"
        path.write_text("\n".join(lines) + "\n")
        p = str(path).encode()
        h_, w_, nz_ = C.c_uint(), C.c_uint(), C.c_uint()
        handle = ref.ref_sm_load_matrix_market(p, C.byref(h_), C.byref(w_), C.byref(nz_))
        assert handle
        rh, rw, rcp, rri, rva = ref_csc(ref, handle)
        gh, gw, gnz = C.c_uint(), C.c_uint(), C.c_uint()
        assert lib.smk_load_matrix_market(p, C.byref(gh), C.byref(gw), C.byref(gnz), None, None, None) == 1
        assert (gh.value, gw.value, gnz.value) == (rh, rw, len(rri))
        cp = np.zeros(rw + 1, dtype=np.uint32)
        ri = np.zeros(len(rri), dtype=np.uint32)
        va = np.zeros(len(rri))
        assert lib.smk_load_matrix_market(p, C.byref(gh), C.byref(gw), C.byref(gnz), up(cp), up(ri), dp(va)) == 1
        assert np.array_equal(cp, rcp) and np.array_equal(ri, rri) and np.array_equal(va, rva)
        ref.ref_sm_free(handle)
