"""pysmallk's ``Flatclust`` and ``Hierclust`` classes (pysmallk/interface/smallk_lib.pyx:924-1420) on the
MI355X library: same method names, keyword arguments and defaults (``load_matrix``, ``load_dictionary``,
``cluster``, ``get_top_indices``, ``get_assignments``, ``get_top_terms``, ``write_output``, ``parser``,
``finalize``).  Python 3; the numeric work is ``smallk_amd.flatclust`` / ``smallk_amd.hierclust``
(C ABI, GPU only, no CPU fallback).  New keyword-only extras: ``seed`` and ``storage``.

SURVEY.md section 2 lists pysmallk's clustering classes as out of scope for the hot path (#21): they get no further work, but
they stay importable where rounds 1-3 had them (``smallk_amd.pyclust``, ``from smallk_amd import Hierclust, Flatclust`` -- loaded
lazily by the package); ``examples/pyclust.py`` re-exports them for the example scripts.
"""
from __future__ import annotations

import argparse
import ctypes as C
import time

import numpy as np

from smallk_amd import _lib as L
from smallk_amd import flatclust as _flat
from smallk_amd import hierclust as _hier
from smallk_amd.solver import initialize, is_initialized, finalize, uniform_host, load_matrix_market


def _load_csv(path):
    h, w = C.c_uint(0), C.c_uint(0)
    d = C.c_double(0)
    if L.lib().smk_load_csv(str(path).encode(), C.byref(d), 0, C.byref(h), C.byref(w)) == 0:
        raise RuntimeError(f"load failed for file {path}")
    buf = np.zeros((h.value, w.value), order="F")
    if L.lib().smk_load_csv(str(path).encode(), buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size, C.byref(h),
                            C.byref(w)) != 1:
        raise RuntimeError(f"load failed for file {path}")
    return buf


class Clustering:
    """Common part (smallk_lib.pyx:924-1078): matrix / dictionary loading and the result getters."""

    def __init__(self):
        if not is_initialized():
            initialize(-1)
        self.dictionary = np.array([])
        self._sparse = False
        self._A = None
        self.height = self.width = self.k = 0
        self.maxterms = 5
        self.term_indices = []
        self.assignments_flat = []
        self.probabilities = []
        self.w = self.h = None
        self.flat = 0

    def finalize(self):
        finalize()

    # :982-995 -- filepath= (.mtx sparse / .csv dense) | sparse buffers | buffer,height,width | matrix=ndarray
    def load_matrix(self, filepath="", height=0, width=0, delim="", buffer=(), nz=0, row_indices=(), col_offsets=(),
                    matrix=(), column_major=False, sparse_matrix=None):
        import scipy.sparse as sp
        if sparse_matrix is not None:
            self._A, self._sparse = sparse_matrix.tocsc(), True
        elif filepath != "":
            if str(filepath).lower().endswith(".mtx"):
                d, ri, cp, shape = load_matrix_market(filepath)
                self._A, self._sparse = sp.csc_matrix((d, ri.astype(np.int64), cp.astype(np.int64)), shape=shape), True
            else:
                self._A, self._sparse = _load_csv(filepath), False
        elif len(row_indices) > 0 and len(col_offsets) > 0:
            self._A = sp.csc_matrix((np.asarray(buffer, dtype=np.float64), np.asarray(row_indices, dtype=np.int64),
                                     np.asarray(col_offsets, dtype=np.int64)), shape=(height, width))
            self._sparse = True
        elif len(buffer) > 0 and height and width:
            self._A = np.asarray(buffer, dtype=np.float64).reshape(width, height).T      # column-major buffer
            self._sparse = False
        elif len(matrix) > 0:
            self._A, self._sparse = np.asarray(matrix, dtype=np.float64), False
        else:
            raise ValueError("load_matrix: no matrix given")
        self.height, self.width = self._A.shape

    # :1001-1010
    def load_dictionary(self, filepath="", dictionary=()):
        if filepath != "":
            with open(filepath) as f:
                terms = f.read().split("\n")
                terms.pop()
            self.dictionary = np.array(terms)
        elif len(dictionary) > 0:
            self.dictionary = np.array(list(dictionary))
        else:
            print("Error: Invalid dictionary.")

    def get_top_indices(self):
        return self.term_indices

    def get_assignments(self):
        return self.assignments_flat

    def get_top_terms(self, filepath="", dictionary=()):
        terms = list(dictionary) if len(dictionary) else list(self.dictionary)
        if filepath != "":
            with open(filepath) as f:
                terms = f.read().split("\n")
                terms.pop()
        return [terms[i] for i in self.get_top_indices()]

    def _flat_post(self):
        self.assignments_flat = _flat.compute_assignments(self.h)
        self.probabilities = _flat.compute_fuzzy_assignments(self.h)
        self.term_indices = _flat.top_terms(self.w, self.maxterms)

    @staticmethod
    def _names(assignfile, fuzzyfile, treefile, k, outdir, fmt):
        ext = ".xml" if fmt == "XML" else ".json"
        tree = outdir + treefile if ext[1:] in treefile else outdir + treefile + "_" + str(k) + ext
        assign = outdir + assignfile if "csv" in assignfile else outdir + assignfile + "_" + str(k) + ".csv"
        fuzzy = outdir + fuzzyfile if "csv" in fuzzyfile else outdir + fuzzyfile + "_" + str(k) + ".csv"
        return assign, fuzzy, tree

    def _write_flat(self, assign, fuzzy, result, fmt):
        res = _flat.FlatResult(0, self.w, self.h, 0, self.assignments_flat, self.probabilities, self.term_indices,
                               self.maxterms)
        return res.write_output(assign, fuzzy, result, list(self.dictionary), fmt)


class Flatclust(Clustering):
    """smallk_lib.pyx:1080-1238"""

    def parser(self):
        p = argparse.ArgumentParser()
        p.add_argument("--matrixfile", action="store", required=True, metavar="matrixfile")
        p.add_argument("--dictfile", action="store", required=True, metavar="dictfile")
        p.add_argument("--clusters", action="store", required=True, metavar="clusters", type=int)
        p.add_argument("--algorithm", action="store", required=False, metavar="algorithm", default="BPP",
                       choices=["HALS", "RANK2", "BPP"])
        p.add_argument("--infile_W", action="store", required=False, metavar="infile_W", default="")
        p.add_argument("--infile_H", action="store", required=False, metavar="infile_H", default="")
        p.add_argument("--tol", action="store", required=False, metavar="tol", type=float, default=0.0001)
        p.add_argument("--outdir", action="store", required=False, metavar="outdir", default="")
        p.add_argument("--miniter", action="store", required=False, metavar="miniter", type=int, default=5)
        p.add_argument("--maxiter", action="store", required=False, metavar="maxiter", type=int, default=5000)
        p.add_argument("--maxterms", action="store", required=False, metavar="maxterms", type=int, default=5)
        p.add_argument("--maxthreads", action="store", required=False, metavar="maxthreads", type=int, default=8)
        p.add_argument("--verbose", action="store", required=False, metavar="verbose", default=True)
        p.add_argument("--format", action="store", required=False, metavar="format", default="XML")
        p.add_argument("--assignfile", action="store", required=False, metavar="assignfile", default="assignments")
        p.add_argument("--treefile", action="store", required=False, metavar="treefile", default="tree")
        p.add_argument("--fuzzyfile", action="store", required=False, metavar="fuzzyfile", default="assignments_fuzzy")
        return p.parse_args()

    def cluster(self, k, infile_W="", infile_H="", algorithm="BPP", maxterms=5, verbose=True, min_iter=5,
                max_iter=5000, max_threads=8, tol=0.0001, *, seed=None, storage="f32"):
        if algorithm.upper() == "RANK2":
            k = 2
        self.k, self.maxterms = k, maxterms
        seed = int(time.time()) if seed is None else int(seed)
        W0 = _load_csv(infile_W) if infile_W else uniform_host(self.height, k, seed)
        H0 = _load_csv(infile_H) if infile_H else uniform_host(k, self.width, seed + 1)
        r = _flat.flatclust(self._A, W0, H0, algorithm.upper(), maxterms=maxterms, storage=storage, min_iter=min_iter,
                            max_iter=max_iter, tol=tol, tolcount=1, prog_est=L.PROG_PG_RATIO, normalize=True,
                            max_threads=max_threads, verbose=bool(verbose))
        if r.result != L.OK:
            raise RuntimeError("NMF solver failure.")
        self.w, self.h, self.iterations = r.W, r.H, r.iteration_count
        self.assignments_flat, self.probabilities, self.term_indices = r.assignments, r.probabilities, r.term_indices

    def write_output(self, assignfile, fuzzyfile, treefile, outdir="./", format="XML"):
        print("Writing output files...")
        assign, fuzzy, tree = self._names(assignfile, fuzzyfile, treefile, self.k, outdir, format)
        return self._write_flat(assign, fuzzy, tree, format)


class Hierclust(Clustering):
    """smallk_lib.pyx:1240-1420"""

    def parser(self):
        p = argparse.ArgumentParser()
        p.add_argument("--matrixfile", action="store", required=True, metavar="matrixfile")
        p.add_argument("--dictfile", action="store", required=True, metavar="dictfile")
        p.add_argument("--clusters", action="store", required=True, metavar="clusters", type=int)
        p.add_argument("--initdir", action="store", required=False, metavar="initdir", default="")
        p.add_argument("--tol", action="store", required=False, metavar="tol", type=float, default=0.0001)
        p.add_argument("--outdir", action="store", required=False, metavar="outdir", default="")
        p.add_argument("--miniter", action="store", required=False, metavar="miniter", type=int, default=5)
        p.add_argument("--maxiter", action="store", required=False, metavar="maxiter", type=int, default=5000)
        p.add_argument("--maxterms", action="store", required=False, metavar="maxterms", type=int, default=5)
        p.add_argument("--maxthreads", action="store", required=False, metavar="maxthreads", type=int, default=8)
        p.add_argument("--unbalanced", action="store", required=False, metavar="unbalanced", type=float, default=0.1)
        p.add_argument("--trial_allowance", action="store", required=False, metavar="trial_allowance", type=int, default=3)
        p.add_argument("--flat", action="store", required=False, metavar="flat", type=int, default=0, choices=[0, 1])
        p.add_argument("--verbose", action="store", required=False, metavar="verbose", default=True)
        p.add_argument("--format", action="store", required=False, metavar="format", default="XML")
        p.add_argument("--treefile", action="store", required=False, metavar="treefile", default="tree")
        p.add_argument("--assignfile", action="store", required=False, metavar="assignfile", default="assignments")
        p.add_argument("--fuzzyfile", action="store", required=False, metavar="fuzzyfile", default="assignments_fuzzy")
        return p.parse_args()

    def cluster(self, k, initdir="", maxterms=5, unbalanced=0.1, trial_allowance=3, verbose=True, flat=0,
                min_iter=5, max_iter=5000, max_threads=8, tol=0.0001, *, seed=None, storage="f32"):
        self.k, self.flat, self.maxterms = k, int(flat), maxterms
        if initdir and not initdir.endswith("/"):
            initdir += "/"
        self.tree = _hier.hier_nmf2(self._A, int(k), seed=int(time.time()) if seed is None else int(seed), initdir=initdir,
                                    storage=storage, tol=float(tol), min_iter=min_iter, max_iter=max_iter,
                                    maxterms=int(maxterms), unbalanced=float(unbalanced),
                                    trial_allowance=int(trial_allowance), verbose=bool(verbose), flat=bool(flat))
        self.nmf_count, self.max_count = self.tree.nmf_count, self.tree.max_count
        if self.flat == 1:
            self.w, self.h = self.tree.flat_factors()
            self._flat_post()

    def get_top_indices(self):
        if self.flat == 1:
            return self.term_indices
        print("ERROR: To get top terms indices, rerun hierarchical clusting with flat=1")
        return []

    def get_assignments(self):
        if self.flat:
            return self.assignments_flat
        return [-1 if int(a) == 4294967295 else int(a) for a in self.tree.get_assignments()]

    def write_output(self, assignfile, treefile, fuzzyfile, outdir="./", format="XML"):
        print("Writing output files...")
        assign, fuzzy, tree = self._names(assignfile, fuzzyfile, treefile, self.k, outdir, format)
        self.tree.write_assignments(assign)
        self.tree.write(tree, list(self.dictionary), format)
        if self.flat == 1:
            # as in the reference (:1395-1397) the flat results are written to the SAME paths afterwards
            self._write_flat(assign, fuzzy, tree, format)
