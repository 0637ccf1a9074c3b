"""examples/cluster_documents.py -- hierarchical + flat clustering of a term-document matrix with the
pysmallk-style classes (needs an MI355X).

    python examples/cluster_documents.py [matrix.mtx dictionary.txt]

Without arguments a small synthetic corpus with planted topics is generated."""
import os
import sys
import tempfile

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smallk_amd import SmallkAPI  # noqa: E402
from pyclust import Hierclust  # noqa: E402  (examples/pyclust.py)


def synthetic(m=2000, n=3000, topics=8, seed=0):
    rng = np.random.default_rng(seed)
    W = np.zeros((m, topics))
    for c in range(topics):
        idx = rng.choice(m, size=m // topics, replace=False)
        W[idx, c] = rng.random(len(idx)) + 0.2
    labels = rng.integers(0, topics, size=n)
    H = np.zeros((topics, n))
    H[labels, np.arange(n)] = rng.random(n) + 0.5
    A = W @ H + 0.02 * rng.random((m, n))
    A[A < 0.25] = 0.0
    return sp.csc_matrix(A), [f"term{i:04d}" for i in range(m)]


def main():
    if len(sys.argv) == 3:
        h = Hierclust()
        h.load_matrix(filepath=sys.argv[1])
        h.load_dictionary(filepath=sys.argv[2])
    else:
        A, dictionary = synthetic()
        h = Hierclust()
        h.load_matrix(sparse_matrix=A)
        h.load_dictionary(dictionary=dictionary)
    h.cluster(8, maxterms=5, flat=1, verbose=False, seed=1)
    out = tempfile.mkdtemp(prefix="smallk_amd_") + "/"
    h.write_output("assignments", "tree", "assignments_fuzzy", outdir=out, format="JSON")
    labels = np.asarray(h.get_assignments())
    print("documents per flat cluster:", np.bincount(labels, minlength=8))
    terms = h.get_top_terms()
    for c in range(8):
        print(f"cluster {c}: " + ", ".join(terms[5 * c:5 * c + 5]))
    print("files written to", out, sorted(os.listdir(out)))

    # the SmallkAPI facade on the same data: plain NMF with BPP
    api = SmallkAPI()
    api.seed_rng(1)              # initial factors are random; without a seed they come from the clock
    if len(sys.argv) == 3:
        api.load_matrix(filepath=sys.argv[1])
    else:
        api.load_matrix(height=A.shape[0], width=A.shape[1], nz=A.nnz, buffer=A.data, row_indices=A.indices,
                        col_offsets=A.indptr)
    api.nmf(8, "BPP", outdir=out)
    print("NMF: W", api.get_W().shape, "H", api.get_H().shape, "iterations", api.get_iteration_count())


if __name__ == "__main__":
    main()
