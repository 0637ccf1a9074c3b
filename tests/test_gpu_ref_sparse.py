"""Resident sparse matrices against the reference's own SparseMatrix code (oracle/_ref/libref_sparse.so):
the CSC and the CSC of the transpose that smk_matrix_create_sparse leaves in HBM, and the column subsets
assembled on the device (sparse_subset.hip), entry for entry."""
import ctypes as C
import os

import numpy as np
import pytest

from test_ref_sparse import REF, ref, ref_csc, random_csc, up, dp   # noqa: F401  (fixture + helpers)

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libref_sparse.so not built")]


def download(l, handle, transposed, ncols):
    nnz = l.smk_matrix_nnz(handle)
    cp = np.zeros(ncols + 1, dtype=np.uint32)
    ri = np.zeros(max(nnz, 1), dtype=np.uint32)
    va = np.zeros(max(nnz, 1))
    assert l.smk_matrix_download_csc(handle, transposed, up(cp), up(ri), dp(va)) == 0
    return cp, ri[:nnz], va[:nnz]


@pytest.mark.parametrize("m,n,density,pick", [(300, 517, 0.02, 140), (5000, 3000, 0.002, 1200), (64, 40, 0.3, 40),
                                               (20000, 9000, 0.0005, 4000)])
def test_resident_csc_transpose_and_device_subset_match_reference(gpu, ref, m, n, density, pick):
    from smallk_amd import _lib as L
    l = L.lib()
    rng = np.random.default_rng(m + n)
    cp, ri, va = random_csc(rng, m, n, density)
    nz = len(ri)
    h = ref.ref_sm_from_csc(m, n, nz, up(cp), up(ri), dp(va))
    src = C.c_void_p()
    L.check(l.smk_matrix_create_sparse(C.byref(src), m, n, 0, n, nz, up(cp), up(ri), dp(va)), "create_sparse")
    # resident CSC == input, resident CSC of A' == reference Transpose
    gcp, gri, gva = download(l, src, 0, n)
    assert np.array_equal(gcp, cp) and np.array_equal(gri, ri) and np.array_equal(gva, va)
    t = ref.ref_sm_transpose(h)
    _, _, tcp, tri, tva = ref_csc(ref, t)
    gcp, gri, gva = download(l, src, 1, m)
    assert np.array_equal(gcp, tcp) and np.array_equal(gri, tri) and np.array_equal(gva, tva)
    ref.ref_sm_free(t)

    nonempty = np.nonzero(np.diff(cp.astype(np.int64)) > 0)[0]
    for trial in range(3):
        cols = np.sort(rng.choice(n, size=pick, replace=False)).astype(np.uint32)       # device cut: increasing lists
        if not np.isin(cols, nonempty).any():
            cols[0] = nonempty[0]
            cols.sort()
        n2o = np.zeros(m, dtype=np.uint32)
        nh = C.c_uint()
        rsub = ref.ref_sm_submatrix_cols_compact(h, up(cols), len(cols), None, up(n2o), C.byref(nh))
        assert rsub
        sh, sw, scp, sri, sva = ref_csc(ref, rsub)
        sub = C.c_void_p()
        rows = np.zeros(m, dtype=np.uint32)
        gh = C.c_int64()
        L.check(l.smk_matrix_gather_cols(src, up(cols), len(cols), C.byref(sub), up(rows), C.byref(gh)), "gather")
        assert gh.value == sh == l.smk_matrix_height(sub) and np.array_equal(rows[:sh], n2o[:sh])
        gcp, gri, gva = download(l, sub, 0, len(cols))
        assert np.array_equal(gcp, scp) and np.array_equal(gri, sri) and np.array_equal(gva, sva)
        # and the transposed CSC of the node == reference Transpose of the reference submatrix
        rt = ref.ref_sm_transpose(rsub)
        _, _, tcp, tri, tva = ref_csc(ref, rt)
        gcp, gri, gva = download(l, sub, 1, sh)
        assert np.array_equal(gcp, tcp) and np.array_equal(gri, tri) and np.array_equal(gva, tva)
        ref.ref_sm_free(rt)
        ref.ref_sm_free(rsub)
        l.smk_matrix_destroy(sub)
    l.smk_matrix_destroy(src)
    ref.ref_sm_free(h)
