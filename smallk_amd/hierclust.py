"""HierNMF2 host mirror: ``Clust`` / ``ClustSparse`` + ``Tree<T>`` behind the C ABI.

Reference surface: hierclust/include/clust.hpp:27-58 (ClustOptions, ClustStats, Clust, ClustSparse),
hierclust/include/tree.hpp (Tree<T>), pysmallk's ``TreeResults`` (smallk_lib.pyx:399-420: ``write``,
``write_assignments``, ``get_assignments``).  All numeric work happens in libsmallk_amd.so on the
GPU; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L

NONE = 0xFFFFFFFF


@dataclass
class TreeNode:
    priority: float
    parent: int
    left: int
    right: int
    is_valid: bool
    is_left_child: bool
    is_leaf: bool
    docs: np.ndarray
    topic_vector: np.ndarray
    term_indices: list


class TreeResults:
    """Owns an ``smk_tree``; mirrors pysmallk's TreeResults plus read access to the nodes."""

    def __init__(self, handle, stats):
        self._h = handle
        self.nmf_count, self.max_count = stats.nmf_count, stats.max_count

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            L.lib().smk_tree_destroy(h)

    @property
    def node_count(self):
        return L.lib().smk_tree_node_count(self._h)

    def node(self, q) -> TreeNode:
        l = L.lib()
        info = L.TreeNodeInfo()
        L.check(l.smk_tree_get_node(self._h, q, C.byref(info)), "smk_tree_get_node")
        docs = np.zeros(max(info.doc_count, 1), dtype=np.uint32)
        l.smk_tree_node_docs(self._h, q, docs.ctypes.data_as(C.POINTER(C.c_uint)))
        topic = np.zeros(l.smk_tree_term_count(self._h))
        l.smk_tree_node_topic(self._h, q, topic.ctypes.data_as(C.POINTER(C.c_double)))
        terms = (C.c_int * 4096)()
        nt = l.smk_tree_node_terms(self._h, q, terms)
        return TreeNode(info.priority, info.parent, info.left_child, info.right_child, bool(info.is_valid),
                        bool(info.is_left_child), bool(info.is_leaf), docs[:info.doc_count], topic,
                        [terms[i] for i in range(nt)])

    @property
    def nodes(self):
        return [self.node(q) for q in range(self.node_count)]

    def get_assignments(self):
        n = L.lib().smk_tree_doc_count(self._h)
        out = np.zeros(n, dtype=np.uint32)
        L.lib().smk_tree_assignments(self._h, out.ctypes.data_as(C.POINTER(C.c_uint)))
        return out

    def get_outliers(self):
        cnt = L.lib().smk_tree_outliers(self._h, None)
        out = np.zeros(max(cnt, 1), dtype=np.uint32)
        L.lib().smk_tree_outliers(self._h, out.ctypes.data_as(C.POINTER(C.c_uint)))
        return out[:cnt]

    def flat_factors(self):
        """(W m x k, H k x n) of the flat clustering run with ``flat=True`` (ClustFlat)."""
        l = L.lib()
        m, n = l.smk_tree_term_count(self._h), l.smk_tree_doc_count(self._h)
        k = sum(1 for q in range(self.node_count) if self.node(q).is_leaf)
        W = np.zeros((m, k), order="F")
        H = np.zeros((k, n), order="F")
        L.check(l.smk_tree_flat_factors(self._h, W.ctypes.data_as(C.POINTER(C.c_double)), m,
                                        H.ctypes.data_as(C.POINTER(C.c_double)), k), "smk_tree_flat_factors")
        return W, H

    def write_assignments(self, filepath):
        return L.lib().smk_tree_write_assignments(self._h, str(filepath).encode()) == L.OK

    def write(self, filepath, dictionary, form="JSON"):
        terms = (C.c_char_p * len(dictionary))(*[str(t).encode() for t in dictionary])
        fmt = 0 if str(form).upper() == "XML" else 1
        return L.lib().smk_tree_write(self._h, str(filepath).encode(), fmt, terms, len(dictionary)) == L.OK


def make_clust_options(m, n, num_clusters, *, tol=1e-4, min_iter=5, max_iter=5000, maxterms=5, unbalanced=0.1,
                       trial_allowance=3, verbose=False, flat=False, prog_est=L.PROG_PG_RATIO):
    """The option set smallk::HierNmf2 uses (smallk/src/smallk.cpp:755-772)."""
    o = L.ClustOptions()
    o.nmf = L.Options(tol, L.ALG_RANK2, prog_est, m, n, 2, min_iter, max_iter, 1, 1, 0, 1)
    o.maxterms, o.unbalanced, o.trial_allowance = maxterms, unbalanced, trial_allowance
    o.num_clusters, o.verbose, o.flat = num_clusters, int(verbose), int(flat)
    return o


def hier_nmf2(A, num_clusters, *, seed=0, draws=0, initdir="", storage="f32", **kw) -> TreeResults:
    """Clust (dense ndarray) / ClustSparse (scipy.sparse matrix) -> TreeResults.  A may also be a
    ``DenseMatrix`` / ``SparseMatrix`` that is already resident in HBM (no upload, ``storage`` ignored)."""
    l = L.lib()
    resident = hasattr(A, "_h") and hasattr(A, "height")       # a DenseMatrix / SparseMatrix already in HBM
    m, n = (A.height, A.ncols) if resident else A.shape
    o = make_clust_options(m, n, num_clusters, **kw)
    tree = C.c_void_p()
    stats = L.ClustStats()
    dr = C.c_uint64(draws)
    idir = initdir.encode() if initdir else None
    if resident:
        rc = l.smk_clust_resident(C.byref(o), A._h, seed, C.byref(dr), idir, C.byref(tree), C.byref(stats))
    elif hasattr(A, "tocsc"):
        a = A.tocsc()
        a.sort_indices()
        co = np.ascontiguousarray(a.indptr, dtype=np.uint32)
        ri = np.ascontiguousarray(a.indices, dtype=np.uint32)
        va = np.ascontiguousarray(a.data, dtype=np.float64)
        rc = l.smk_clust_sparse(C.byref(o), a.nnz, co.ctypes.data_as(C.POINTER(C.c_uint)),
                                ri.ctypes.data_as(C.POINTER(C.c_uint)), va.ctypes.data_as(C.POINTER(C.c_double)),
                                seed, C.byref(dr), idir, C.byref(tree), C.byref(stats))
    else:
        a = np.asfortranarray(A, dtype=np.float64)
        st = L.STORE_BF16 if str(storage).lower() == "bf16" else L.STORE_F32
        rc = l.smk_clust_dense(C.byref(o), a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], st, seed,
                               C.byref(dr), idir, C.byref(tree), C.byref(stats))
    if rc != L.OK and tree:            # FLATCLUST_FAILURE hands the tree back; this wrapper raises instead
        l.smk_tree_destroy(tree)
    L.check(rc, "smk_clust")
    res = TreeResults(tree, stats)
    res.draws = dr.value
    return res


def priority(w_parent, w_child) -> float:
    """compute_priority (clust_hier_util.hpp:105-173); host-side, no GPU needed."""
    wp = np.ascontiguousarray(w_parent, dtype=np.float64).ravel()
    wc = np.asfortranarray(w_child, dtype=np.float64)
    return L.lib().smk_clust_priority(wp.ctypes.data_as(C.POINTER(C.c_double)),
                                      wc.ctypes.data_as(C.POINTER(C.c_double)), len(wp))
