"""oracle.flatclust -- CPU restatement of the flat-clustering pieces.  TEST INFRASTRUCTURE ONLY.

  nnls_hals                 common/include/nnls.hpp:249-316 (UpdateH_Hals: nmf_solver_hals.hpp:26-62,
                            ProjectedGradientNorm: projected_gradient.hpp:94-121)
  compute_assignments       common/include/assignments.hpp:72-113
  compute_fuzzy_assignments common/include/assignments.hpp:32-69
  top_terms                 common/include/terms.hpp:62-108
  assignments_text / fuzzy_text / results_text
                            common/src/assignments.cpp:23-70, flat_clust_output.cpp:56-141,
                            flatclust_{json,xml}_writer.cpp

Pinning: the text writers and the two assignment routines are compared with the reference's own
code compiled into oracle/_ref/libref_flat.so (tests/test_flatclust.py).  nnls_hals is "parity
unpinned" like the solvers (Elemental), checked by its fixed-point property instead.
"""
from __future__ import annotations

import numpy as np

from . import nmf, OK


def nnls_hals(A, W, H0, tol, max_iter):
    """Returns (success, W, H, iterations); W, H normalised on success (NormalizeAndScale)."""
    if hasattr(A, "toarray"):
        A = A.toarray()
    A = np.asarray(A, dtype=np.float64)
    W = np.array(W, dtype=np.float64, order="F")
    H = np.array(H0, dtype=np.float64, order="F")
    k = W.shape[1]
    WtW = W.T @ W
    WtA = W.T @ A
    pg0 = 0.0
    for i in range(max_iter):
        for r in range(k):
            with np.errstate(divide="ignore", invalid="ignore"):
                h = H[r, :] + (WtA[r, :] - WtW[r, :] @ H) / WtW[r, r]
            h[np.isnan(h) | (h < 0)] = 0.0
            H[r, :] = h
        grad = WtW @ H - WtA
        pg = float(np.sqrt(np.sum(grad[(grad < 0) | (H > 0)] ** 2)))
        if i == 0:
            pg0 = pg
            continue
        if pg < tol * pg0:
            nu = np.sqrt(np.sum(W * W, axis=0))          # normalize.hpp:118-140
            return True, W / nu, H * nu[:, None], i + 1
    return False, W, H, max_iter


def compute_assignments(H):
    H = np.asarray(H)
    return np.argmax(H, axis=0).astype(np.uint32)        # first maximum, like the strict '>' scan


def compute_fuzzy_assignments(H):
    H = np.asarray(H, dtype=np.float64)
    s = np.zeros(H.shape[1])
    for r in range(H.shape[0]):                          # sequential sum, same order as the reference
        s = s + H[r, :]
    return (H * (1.0 / s)).astype(np.float32)            # k x n; file order is column by column


def top_terms(W, maxterms):
    W = np.asarray(W)
    m, k = W.shape
    out = np.zeros((k, maxterms), dtype=np.int32)
    for c in range(k):
        order = np.lexsort((np.arange(m), -W[:, c]))
        cnt = min(maxterms, m)
        out[c, :cnt] = order[:cnt]
    return out.ravel()


def assignments_text(labels):
    return ",".join(str(int(x)) for x in labels) + "\n"


def fuzzy_text(P):
    P = np.asarray(P)
    return "".join(",".join(f"{float(P[r, c]):.3e}" for r in range(P.shape[0])) + "\n" for c in range(P.shape[1]))


def results_text(labels, term_indices, dictionary, fmt, maxterms, num_docs, num_clusters):
    S4 = "    "
    S8, S12, S16 = S4 * 2, S4 * 3, S4 * 4
    counts = {}
    for x in labels:
        counts[int(x)] = counts.get(int(x), 0) + 1
    o = []
    json = fmt.upper() == "JSON"
    if json:
        o.append("{\n" + S4 + f"\"doc_count\": {num_docs},\n" + S4 + "\"nodes\": [\n")
    else:
        o.append("<?xml version=\"1.0\"?>\n" + f"<DataSet id=\"{num_docs}\">\n")
    for i in range(num_clusters):
        terms = [dictionary[int(t)] for t in term_indices[i * maxterms:(i + 1) * maxterms]] if i in counts else None
        if json:
            if i:
                o.append(",\n")
            o.append(S8 + "{\n" + S12 + f"\"id\": {i},\n" + S12 + f"\"doc_count\": {counts.get(i, 0)},\n")
            if terms:
                o.append(S12 + "\"top_terms\": [\n" + ",\n".join(S16 + f"\"{t}\"" for t in terms) + "\n" + S12 + "]\n")
            o.append(S8 + "}")
        else:
            o.append(S4 + f"<node id=\"{i}\">\n" + S8 + f"<doc_count>{counts.get(i, 0)}</doc_count>\n")
            if terms is not None:
                o.append(S8 + "<top_terms>\n" + "".join(S12 + f"<term name=\"{t}\"/>\n" for t in terms) + S8 + "</top_terms>\n")
            o.append(S4 + "</node>\n")
    o.append("\n" + S4 + "]\n}\n" if json else "</DataSet>\n")
    return "".join(o)


def flatclust(A, W0, H0, algorithm, **kw):
    """FlatClust (flatclust/src/flat_clust.cpp:118-190) = NmfSolve with HALS / RANK2 / BPP."""
    if algorithm not in ("HALS", "RANK2", "BPP"):
        raise ValueError("unknown NMF algorithm")
    if hasattr(A, "toarray"):
        A = A.toarray()
    return nmf(A, W0, H0, algorithm, **kw)
