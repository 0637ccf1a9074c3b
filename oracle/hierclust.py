"""oracle.hierclust -- CPU restatement of HierNMF2 (rank-2 hierarchical clustering).

TEST INFRASTRUCTURE ONLY (same rule as the rest of ``oracle/``).  PARITY UNPINNED for the
tree search itself: the reference ships no tree/assignment fixtures in this checkout (its
test scripts read an external data directory) and the code cannot be compiled here
(Elemental).  The file writers ARE pinned: ``oracle/_ref`` compiles the reference's own
``hierclust_{json,xml}_writer.cpp`` and ``tests/test_hierclust.py`` compares bytes.

Follows, function by function:
  ClustHier        hierclust/include/clust_hier_generic.hpp:67-196
  TrialSplit       hierclust/include/clust_hier_generic.hpp:203-327
  ActualSplit      hierclust/include/clust_hier_generic.hpp:383-499
  compute_priority hierclust/include/clust_hier_util.hpp:105-173 (+ NDCG_part :50-99, ordered :25-47)
  Tree<T>          hierclust/include/tree.hpp (Init :120, MinMaxLeafPriorities :147, SplitRoot :173,
                   Split :214, ComputeTopTerms :282, ComputeAssignments :302, WriteAssignments :388)
  SetDiff          hierclust/include/setdiff.hpp:23-46
  SubMatrixColsCompact  common/include/sparse_matrix_impl.hpp:479-590, dense_matrix_impl.hpp:224-281

Every rank-2 factorisation is ``oracle.nmf(..., "RANK2")`` (the C restatement); the random
initialisers are the counter-based generator shared with the device library, drawn in the
reference's order (W then H per attempt, clust_hier_util.hpp:196-203).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import fill_uniform, nmf, nmf_sparse, PG_RATIO, OK

NONE = 0xFFFFFFFF
SEED_STRIDE = 0x9E37          # smallk_amd/csrc/facade.cpp RandomMatrix(): seed + stride * (++draws)


@dataclass
class Node:
    priority: float = 0.0
    parent: int = NONE
    left: int = NONE
    right: int = NONE
    is_valid: bool = False
    is_left_child: bool = False
    topic_vector: np.ndarray = None
    term_indices: list = field(default_factory=list)
    docs: list = field(default_factory=list)


class Tree:
    def __init__(self, num_clusters, node_count, term_count, doc_count):
        self.total_docs = doc_count
        self.nodes = [Node(topic_vector=np.zeros(term_count)) for _ in range(node_count)]
        self.is_leaf = [False] * node_count
        self.active = 0
        self.index0 = self.index1 = 0
        self.outliers = []
        self.assignments = []
        self.leaf_doc_count = 0

    def min_max_leaf_priorities(self):
        mn, mx, idx = np.finfo(np.float64).max, np.finfo(np.float64).min, 0
        for q, leaf in enumerate(self.is_leaf):
            if not leaf:
                continue
            p = self.nodes[q].priority
            if p > 0 and p < mn:
                mn = p
            if p > mx:
                mx, idx = p, q
        return mn, mx, idx

    def _open(self, parent):
        for i, left in ((self.index0, True), (self.index1, False)):
            nd = self.nodes[i]
            nd.parent, nd.left, nd.right, nd.is_valid, nd.is_left_child = parent, NONE, NONE, True, left
            self.is_leaf[i] = True

    def split_root(self, W, H):
        self.index0, self.index1 = 0, 1
        self._open(NONE)
        self.active += 2
        for c in range(H.shape[1]):
            (self.nodes[0] if H[0, c] > H[1, c] else self.nodes[1]).docs.append(c)
        self._topics(W)

    def split(self, node_index, W, H):
        self.index0, self.index1 = self.active, self.active + 1
        self.active += 2
        self.nodes[node_index].left, self.nodes[node_index].right = self.index0, self.index1
        self.is_leaf[node_index] = False
        self._open(node_index)
        src = self.nodes[node_index].docs
        for c in range(H.shape[1]):
            (self.nodes[self.index0] if H[0, c] > H[1, c] else self.nodes[self.index1]).docs.append(src[c])
        self._topics(W)

    def _topics(self, W):
        self.nodes[self.index0].topic_vector = W[:, 0].copy()
        self.nodes[self.index1].topic_vector = W[:, 1].copy()

    def compute_top_terms(self, max_terms):
        for nd in self.nodes:
            if not nd.is_valid:
                continue
            # std::sort by value descending (terms.hpp:44-45); ties resolved by index here
            order = np.lexsort((np.arange(len(nd.topic_vector)), -nd.topic_vector))
            terms = [0] * max_terms
            for q in range(min(max_terms, len(order))):
                terms[q] = int(order[q])
            nd.term_indices = terms

    def compute_assignments(self):
        self.assignments = [NONE] * self.total_docs
        self.leaf_doc_count = 0
        for q, nd in enumerate(self.nodes):
            if not self.is_leaf[q]:
                continue
            self.leaf_doc_count += len(nd.docs)
            for d in nd.docs:
                self.assignments[d] = q
        self.outliers = [q for q, a in enumerate(self.assignments) if a == NONE]

    def assignments_text(self) -> str:
        """Tree::WriteAssignments, tree.hpp:388-423."""
        out = [str(self.assignments[0])]
        for a in self.assignments[1:]:
            out.append("," + ("-1" if a == NONE else str(a)))
        s = "".join(out) + "\n\n"
        if self.outliers:
            s += ",".join(str(o) for o in self.outliers) + "\n"
        return s


def _ordered(v):
    v = np.asarray(v)
    return np.lexsort((np.arange(len(v)), v))


def _desc_ordered(v):
    v = np.asarray(v)
    return np.lexsort((np.arange(len(v)), -v))


def _seq_sum(x):
    s = 0.0
    for t in x:
        s += t
    return s


def ndcg_part(ground, test, weight, weight_part):
    seq_idx = _ordered(ground)
    twp = np.asarray(weight_part)[seq_idx]
    n = len(test)
    disc = np.ones(n)
    disc[1:] = np.log2(np.arange(1, n) + 1.0)
    uncum = twp[np.asarray(test)] / disc
    ideal = np.sort(np.asarray(weight))[::-1] / disc
    return _seq_sum(uncum) / _seq_sum(ideal)


def compute_priority(w_parent, w_child):
    w_parent = np.asarray(w_parent, dtype=np.float64).ravel()
    n = len(w_parent)
    n_part = int(np.count_nonzero(w_parent))
    idx_parent = _desc_ordered(w_parent)
    idx_c1 = _desc_ordered(w_child[:, 0])
    idx_c2 = _desc_ordered(w_child[:, 1])
    if n_part <= 1:
        return -3.0
    weight = np.log(np.arange(n, 0, -1).astype(np.float64))
    zeros = np.nonzero(w_parent[idx_parent] == 0)[0]
    if len(zeros):
        weight[zeros[0]:] = 1.0
    weight_part = np.zeros(n)
    weight_part[:n_part] = np.log(np.arange(n_part, 0, -1).astype(np.float64))
    idx1, idx2 = _ordered(idx_c1), _ordered(idx_c2)
    max_pos = np.maximum(idx1, idx2)
    discount = np.log((n - max_pos[idx_parent]).astype(np.float64))
    discount[discount == 0] = math.log(2.0)
    weight = weight / discount
    weight_part = weight_part / discount
    return ndcg_part(idx_parent, idx_c1, weight, weight_part) * ndcg_part(idx_parent, idx_c2, weight, weight_part)


def set_diff(a, b):
    out, i = [], 0
    for x in b:
        while a[i] < x:
            out.append(a[i])
            i += 1
        i += 1
    out.extend(a[i:])
    return out


class _Source:
    """A with the reference's two SubMatrixColsCompact behaviours."""

    def __init__(self, A):
        try:
            import scipy.sparse as sp
            self.sparse = sp.issparse(A)
        except ImportError:          # pragma: no cover
            self.sparse = False
        self.A = A.tocsc() if self.sparse else np.asfortranarray(A, dtype=np.float64)
        self.m, self.n = self.A.shape

    DENSE_LIMIT = 1 << 24

    def full(self):
        if self.sparse and self.m * self.n > self.DENSE_LIMIT:
            return self.A
        return self.A.toarray(order="F") if self.sparse else self.A

    def subset(self, cols):
        if not self.sparse:
            return self.A[:, cols], np.arange(self.m)
        sub = self.A[:, cols]
        if sub.nnz == 0 and len(sub.indices) == 0:
            raise ValueError("SparseMatrix::SubMatrixColsCompact: submatrix is the zero matrix")
        used = np.zeros(self.m, dtype=bool)
        used[sub.indices] = True            # structural entries, explicit zeros included
        rows = np.nonzero(used)[0]
        sub = sub[rows, :]
        # small nodes as dense arrays (what the tests always did); large ones stay CSC and are factored by
        # oracle.nmf_sparse -- same driver and solver, the products with A run over the stored entries
        if sub.shape[0] * sub.shape[1] <= self.DENSE_LIMIT:
            return sub.toarray(order="F"), rows
        return sub.tocsc(), rows


@dataclass
class ClustStats:
    nmf_count: int = 0
    max_count: int = 0


class _Init:
    def __init__(self, seed, draws, initializers):
        self.seed, self.draws = seed, draws
        self.files = list(initializers) if initializers is not None else None
        self.counter = 0

    def next_full(self):
        W, H = self.files[self.counter]
        self.counter += 1
        return np.asfortranarray(W, dtype=np.float64), np.asfortranarray(H, dtype=np.float64)

    def draw(self, rows, cols):
        self.draws += 1
        return fill_uniform(rows, cols, (self.seed + SEED_STRIDE * self.draws) & 0xFFFFFFFFFFFFFFFF)

    def random(self, h, w):
        W = self.draw(h, 2)
        H = self.draw(2, w)
        return W, H


def hier_nmf2(A, num_clusters, *, tol=1e-4, min_iter=5, max_iter=5000, maxterms=5, unbalanced=0.1,
              trial_allowance=3, seed=0, draws=0, initializers=None, flat=False):
    """Returns (tree, stats).  ``flat``: also run ClustFlat (clust_flat_generic.hpp:33-74) on the leaf topic
    vectors; the factors land in ``tree.flat_W`` (m x k) / ``tree.flat_H`` (k x n).  ``initializers``: sequence of full-size (W m x 2, H 2 x n) pairs,
    consumed like the reference's Winit_<i>.csv / Hinit_<i>.csv files (clust_hier_util.hpp:206-241)."""
    if num_clusters <= 1:
        raise ValueError("HierNMF2: number of clusters must be >= 2")
    src = _Source(A)
    m, n = src.m, src.n
    init = _Init(seed, draws, initializers)
    stats = ClustStats()
    node_count = 2 * (num_clusters - 1)
    tree = Tree(num_clusters, node_count, m, n)

    def solve(Asub, W0, H0):
        run = nmf if isinstance(Asub, np.ndarray) else nmf_sparse
        r = run(Asub, W0, H0, "RANK2", min_iter=min_iter, max_iter=max_iter, tol=tol, tolcount=1,
                prog_est=PG_RATIO, normalize=True)
        return r.result == OK, r.W, r.H, r.iteration_count

    def factor(Asub, rows, cols):
        for _ in range(3):
            if init.files is not None:
                Wf, Hf = init.next_full()
                W0, H0 = Wf[rows, :], Hf[:, cols]
            else:
                W0, H0 = init.random(Asub.shape[0], Asub.shape[1])
            ok, W, H, it = solve(Asub, W0, H0)
            if ok:
                stats.nmf_count += 1
                if it == max_iter:
                    stats.max_count += 1
                return W, H
        raise RuntimeError("HierNMF2: node factorization failed after three attempts.")

    def actual_split(subset, w_parent):
        if len(subset) <= 3:
            return -1.0, np.zeros((m, 2), order="F"), np.zeros((2, len(subset)), order="F"), [1] * len(subset)
        Asub, rows = src.subset(subset)
        Ws, Hs = factor(np.asfortranarray(Asub) if isinstance(Asub, np.ndarray) else Asub, rows, subset)
        labels = [0 if Hs[0, c] > Hs[1, c] else 1 for c in range(Hs.shape[1])]
        W = np.zeros((m, 2), order="F")
        W[rows, :] = Ws
        pr = -1.0
        if 0 in labels and 1 in labels:
            pr = compute_priority(w_parent, W)
        return pr, W, Hs.copy(order="F"), labels

    def trial_split(node, min_priority):
        subset = node.docs
        backup = list(subset)
        subset_small = []
        trial = 0
        pr = -2.0
        W = H = None
        while trial < trial_allowance:
            pr, W, H, labels = actual_split(subset, node.topic_vector)
            if pr < 0:
                break
            counts = [labels.count(0), labels.count(1)]
            smallest = min(counts)
            if smallest < unbalanced * len(labels):
                lab = 0 if smallest == counts[0] else 1
                subset_small = [subset[q] for q in range(len(labels)) if labels[q] == lab]
                pr_small, _, _, _ = actual_split(subset_small, W[:, lab])
                if pr_small < min_priority:
                    trial += 1
                    if trial < trial_allowance:
                        subset = set_diff(subset, subset_small)
                        node.docs = subset
                else:
                    break
            else:
                break
        if trial == trial_allowance:
            node.docs = backup
            W = np.zeros((m, 2), order="F")
            H = np.zeros((2, len(backup)), order="F")
            pr = -2.0
        if W is None:                       # trial_allowance == 0: the reference asserts here
            raise AssertionError("TrialSplit: no split computed")
        return pr, W, H

    # root (clust_hier_generic.hpp:97-121)
    Afull = src.full()
    W0, H0 = factor(np.asfortranarray(Afull) if isinstance(Afull, np.ndarray) else Afull, np.arange(m), list(range(n)))
    Wbuf, Hbuf = [None] * node_count, [None] * node_count
    for i in range(num_clusters - 1):
        if i == 0:
            min_priority = math.inf
            tree.split_root(W0, H0)
        else:
            min_priority, max_priority, split_index = tree.min_max_leaf_priorities()
            if max_priority < 0:
                break
            tree.split(split_index, Wbuf[split_index], Hbuf[split_index])
        for idx in (tree.index0, tree.index1):
            pr, Wbuf[idx], Hbuf[idx] = trial_split(tree.nodes[idx], min_priority)
            tree.nodes[idx].priority = pr
    tree.compute_top_terms(maxterms)
    tree.compute_assignments()
    tree.flat_W = tree.flat_H = None
    if flat:
        from .flatclust import nnls_hals
        leaves = [q for q in range(node_count) if tree.is_leaf[q]]
        if len(leaves) != num_clusters:              # Tree::FlatclustInitW, tree.hpp:341-385
            raise RuntimeError("Insufficient number of leaf nodes for flat clustering.")
        Wl = np.stack([tree.nodes[q].topic_vector for q in leaves], axis=1)
        for _ in range(3):
            ok, Wf, Hf, _its = nnls_hals(src.full(), Wl, init.draw(num_clusters, n), tol, max_iter)
            if ok:
                tree.flat_W, tree.flat_H = Wf, Hf
                break
        else:
            raise RuntimeError("Flatclust NNLS solver failed after 3 attempts.")
    tree.draws = init.draws
    return tree, stats


# ---- file writers (hierclust_json_writer.cpp / hierclust_xml_writer.cpp) restated as strings ----
def _signed(v):
    return v - (1 << 32) if v >= (1 << 31) else v


def tree_text(tree: Tree, dictionary, fmt: str) -> str:
    S4 = "    "
    S8, S12, S16 = S4 * 2, S4 * 3, S4 * 4
    o = []
    if fmt.upper() == "JSON":
        o.append("{\n" + S4 + f"\"doc_count\": {tree.leaf_doc_count},\n" + S4 + "\"nodes\": [\n")
        for q, nd in enumerate(tree.nodes):
            if q:
                o.append(",\n")
            o.append(S8 + "{\n" + S12 + f"\"id\": {q},\n")
            o.append(S12 + f"\"parent_id\": {_signed(nd.parent)},\n")
            o.append(S12 + f"\"left_child\": {'true' if nd.is_left_child else 'false'},\n")
            o.append(S12 + f"\"left_child_id\": {_signed(nd.left)},\n")
            o.append(S12 + f"\"right_child_id\": {_signed(nd.right)},\n")
            o.append(S12 + f"\"doc_count\": {len(nd.docs)},\n")
            if nd.term_indices:
                o.append(S12 + "\"top_terms\": [\n")
                o.append(",\n".join(S16 + f"\"{dictionary[t]}\"" for t in nd.term_indices) + "\n")
                o.append(S12 + "]\n")
            o.append(S8 + "}")
        o.append("\n" + S4 + "]\n}\n")
    else:
        o.append("<?xml version=\"1.0\"?>\n" + f"<DataSet id=\"{tree.leaf_doc_count}\">\n")
        for q, nd in enumerate(tree.nodes):
            o.append(S4 + f"<node id=\"{q}\">\n")
            o.append(S8 + f"<parent_id>{_signed(nd.parent)}</parent_id>\n")
            o.append(S8 + f"<left_child>{'true' if nd.is_left_child else 'false'}</left_child>\n")
            o.append(S8 + f"<left_child_id>{_signed(nd.left)}</left_child_id>\n")
            o.append(S8 + f"<right_child_id>{_signed(nd.right)}</right_child_id>\n")
            o.append(S8 + f"<doc_count>{len(nd.docs)}</doc_count>\n")
            o.append(S8 + "<top_terms>\n")
            for t in nd.term_indices:
                o.append(S12 + f"<term name=\"{dictionary[t]}\"/>\n")
            o.append(S8 + "</top_terms>\n")
            o.append(S4 + "</node>\n")
        o.append("</DataSet>\n")
    return "".join(o)
