"""Synthetic sparse inputs of the bench workloads and tests (host side, numpy/scipy; the reference's data sets --
smallk_data/reuters.mtx, Makefile:28 -- are not in the tree).

term_document(m, n, nnz, seed): a term-by-document count matrix in the shape of the reference's Reuters example
  (sphinx/source/pages_smallkAPI.rst:58-143: 12411 terms x 7984 documents): document lengths log-normal, terms drawn from a
  Zipf law (exponent 1.0) -- a few terms occur in thousands of documents, most in a handful -- tf weights 1 + log(count), every
  row and column non-empty (an empty row or column makes W'W or HH' singular under block pivoting).
community_graph(n, degree, communities, seed): the symmetric adjacency of tools/c5_hier.py (C5's shape): `communities` planted
  groups, 85 % of the edges inside a group, unit weights (duplicates summed)."""
import numpy as np
import scipy.sparse as sp


def term_document(m, n, nnz, seed=0):
    rng = np.random.default_rng(seed)
    lens = rng.lognormal(mean=0.0, sigma=0.6, size=n)
    lens = np.maximum(1, np.round(lens * (1.25 * nnz / lens.sum()))).astype(np.int64)   # duplicates collapse: draw ~25 % more
    p = 1.0 / np.arange(1, m + 1, dtype=np.float64)
    p /= p.sum()
    cdf = np.cumsum(p)
    total = int(lens.sum())
    terms = np.searchsorted(cdf, rng.random(total), side="right").astype(np.int64)
    terms = np.minimum(terms, m - 1)
    perm = rng.permutation(m)                           # frequent terms are not the first rows
    docs = np.repeat(np.arange(n, dtype=np.int64), lens)
    A = sp.coo_matrix((np.ones(total), (perm[terms], docs)), shape=(m, n)).tocsc()
    A.sum_duplicates()
    A.data = 1.0 + np.log(A.data)
    # every term occurs somewhere, every document has a term
    rows_missing = np.flatnonzero(np.diff(A.tocsr().indptr) == 0)
    if rows_missing.size:
        extra = sp.coo_matrix((np.ones(rows_missing.size), (rows_missing, rng.integers(0, n, rows_missing.size))), shape=(m, n))
        A = (A + extra).tocsc()
    A.sort_indices()
    return A


def community_graph(n, degree=16, communities=16, seed=0):
    rng = np.random.default_rng(seed)
    comm = rng.integers(0, communities, size=n)
    order = np.argsort(comm, kind="stable")
    starts = np.searchsorted(comm[order], np.arange(communities + 1))
    nnz_half = n * degree // 2
    src = rng.integers(0, n, size=nnz_half)
    intra = rng.random(nnz_half) < 0.85
    dst = rng.integers(0, n, size=nnz_half)
    c = comm[src[intra]]
    dst[intra] = order[starts[c] + (rng.random(int(intra.sum())) * (starts[c + 1] - starts[c])).astype(np.int64)]
    A = sp.coo_matrix((np.ones(nnz_half), (src, dst)), shape=(n, n))
    A = (A + A.T).tocsc()
    A.sum_duplicates()
    A.sort_indices()
    return A, comm
