"""Flat clustering host mirror: FlatClust / FlatClustSparse, NnlsHals and the result files.

Reference surface: flatclust/include/flat_clust.hpp (FlatClust, FlatClustSparse), common/include/nnls.hpp:249-316
(NnlsHals), common/include/assignments.hpp, terms.hpp:62-108, common/src/flat_clust_output.cpp.  pysmallk's
``Flatclust`` class (smallk_lib.pyx) exposes the same pieces as ``cluster`` / ``get_top_indices`` /
``get_assignments`` / ``write_output``.  The factorisation runs on the GPU; no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from .solver import make_options, _f, _p, STORAGE, DenseMatrix, SparseMatrix, NmfSolver

_up = C.POINTER(C.c_uint)


@dataclass
class FlatResult:
    result: int
    W: np.ndarray
    H: np.ndarray
    iteration_count: int
    assignments: np.ndarray          # argmax row of each column of H
    probabilities: np.ndarray        # k x n float32, columns sum to 1
    term_indices: np.ndarray         # k * maxterms
    maxterms: int

    def write_output(self, assignfile, fuzzyfile, resultfile, dictionary, form="JSON"):
        """FlatClustWriteResults (flat_clust_output.cpp:56-141)."""
        k, n = self.H.shape
        a = np.ascontiguousarray(self.assignments, dtype=np.uint32)
        p = np.ascontiguousarray(self.probabilities.T, dtype=np.float32)      # file order: document by document
        t = np.ascontiguousarray(self.term_indices, dtype=np.int32)
        d = (C.c_char_p * len(dictionary))(*[str(x).encode() for x in dictionary])
        rc = L.lib().smk_flatclust_write_results(str(assignfile).encode(), str(fuzzyfile).encode(),
                                                 str(resultfile).encode(), a.ctypes.data_as(_up), len(a),
                                                 p.ctypes.data_as(C.POINTER(C.c_float)), d, len(dictionary),
                                                 t.ctypes.data_as(C.POINTER(C.c_int)), len(t),
                                                 0 if str(form).upper() == "XML" else 1, self.maxterms, n, k)
        return rc == L.OK

    def write_to_dir(self, outdir, dictionary, form="JSON"):
        """The file names FlatClustWriteResults(outdir, ...) uses (flat_clust_output.cpp:144-173)."""
        k = self.H.shape[0]
        ext = ".xml" if str(form).upper() == "XML" else ".json"
        return self.write_output(os.path.join(outdir, f"assignments_flat_{k}.csv"),
                                 os.path.join(outdir, f"assignments_fuzzy_{k}.csv"),
                                 os.path.join(outdir, f"clusters_{k}{ext}"), dictionary, form)


def compute_assignments(H):
    H = _f(H)
    out = np.zeros(H.shape[1], dtype=np.uint32)
    L.check(L.lib().smk_compute_assignments(_p(H), H.shape[0], H.shape[0], H.shape[1], out.ctypes.data_as(_up)),
            "smk_compute_assignments")
    return out


def compute_fuzzy_assignments(H):
    H = _f(H)
    out = np.zeros(H.shape, dtype=np.float32, order="F")
    L.check(L.lib().smk_compute_fuzzy_assignments(_p(H), H.shape[0], H.shape[0], H.shape[1],
                                                  out.ctypes.data_as(C.POINTER(C.c_float))),
            "smk_compute_fuzzy_assignments")
    return out


def top_terms(W, maxterms):
    W = _f(W)
    out = np.zeros(W.shape[1] * maxterms, dtype=np.int32)
    L.check(L.lib().smk_top_terms(maxterms, _p(W), W.shape[0], W.shape[0], W.shape[1],
                                  out.ctypes.data_as(C.POINTER(C.c_int))), "smk_top_terms")
    return out


def _post(rc, W, H, iters, maxterms):
    if rc != L.OK:
        return FlatResult(rc, W, H, iters, None, None, None, maxterms)
    return FlatResult(rc, W, H, iters, compute_assignments(H), compute_fuzzy_assignments(H), top_terms(W, maxterms),
                      maxterms)


def flatclust(A, W0, H0, algorithm="HALS", *, maxterms=5, storage="f32", **kw) -> FlatResult:
    """FlatClust (dense ndarray) / FlatClustSparse (scipy.sparse) followed by the post-processing of
    flatclust/src/main.cpp:225-262 (assignments, fuzzy assignments, top terms)."""
    W = _f(W0).copy(order="F")
    H = _f(H0).copy(order="F")
    m, n = A.shape
    k = W.shape[1]
    o = make_options(m, n, k, algorithm, **kw)
    st = L.Stats()
    if hasattr(A, "tocsc"):
        a = A.tocsc()
        d = np.ascontiguousarray(a.data, dtype=np.float64)
        ri = np.ascontiguousarray(a.indices, dtype=np.uint32)
        co = np.ascontiguousarray(a.indptr, dtype=np.uint32)
        rc = L.lib().smk_flatclust_sparse(C.byref(o), m, n, d.size, co.ctypes.data_as(_up), ri.ctypes.data_as(_up),
                                          _p(d), _p(W), m, _p(H), k, C.byref(st))
    else:
        a = _f(A)
        stg = STORAGE[storage] if isinstance(storage, str) else int(storage)
        rc = L.lib().smk_flatclust_dense(C.byref(o), _p(a), m, _p(W), m, _p(H), k, C.byref(st), stg)
    if rc not in (L.OK, L.FAILURE, L.BAD_PARAM, L.NOTINITIALIZED, L.SIZE_TOO_LARGE):
        L.check(rc, "smk_flatclust")
    return _post(rc, W, H, st.iteration_count, maxterms)


def nnls_hals(A, W, H0, *, tol=1e-4, max_iter=5000, verbose=False, storage="f32"):
    """NnlsHals (nnls.hpp:249-316) on a resident matrix.  Returns (result code, W, H, iterations)."""
    mat = SparseMatrix.from_scipy(A) if hasattr(A, "tocsc") else DenseMatrix.from_host(_f(A), storage=storage)
    W = _f(W)
    k = W.shape[1]
    s = NmfSolver(mat, make_options(mat.height, mat.ncols, k, "HALS"))
    s.set_factors(W, H0)
    its = C.c_int(0)
    rc = L.lib().smk_solver_nnls_hals(s._h, tol, int(verbose), max_iter, C.byref(its))
    if rc not in (L.OK, L.FAILURE):
        L.check(rc, "smk_solver_nnls_hals")
    Wn, Hn = s.factors()
    s.close()
    mat.close()
    return rc, Wn, Hn, its.value
