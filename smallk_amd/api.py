"""Python-3 mirror of pysmallk's ``SmallkAPI`` class over the C ABI.

Method names, keyword arguments and defaults follow the reference binding
(pysmallk/interface/smallk_lib.pyx:634-921); like it, every method is a thin
pass-through to the C++ API ``namespace smallk`` (include/smallk.hpp), reached
here through the flat ``smk_api_*`` handles because ctypes cannot call C++.
C++ exceptions surface as ``RuntimeError`` (Cython's ``except +`` does the same
for std::logic_error / std::runtime_error).
"""
from __future__ import annotations

import argparse
import ctypes as C

import numpy as np

from . import _lib as L

# smallk::Algorithm numbering (smallk/include/smallk.hpp:34-40) -- differs from NmfAlgorithm
_ALG = {"MU": 0, "BPP": 1, "HALS": 2, "RANK2": 3}


def _raise_if(status, where):
    if status != 0:
        msg = L.lib().smk_api_last_exception()
        raise RuntimeError(f"{where}: {msg.decode() if msg else 'error'}")


def _b(s) -> bytes:
    return s if isinstance(s, bytes) else str(s).encode()


class SmallkAPI:
    def __init__(self):
        _raise_if(L.lib().smk_api_initialize(), "Initialize")
        if not L.lib().smk_api_is_initialized():
            print("ERROR")
        self._dictionary_loaded = False

    # -- command line parser of the default application (smallk_lib.pyx:646-691) ------------
    def parser(self):
        p = argparse.ArgumentParser(description="Run NMF via python binding")
        p.add_argument("--matrixfile", action="store", required=True, metavar="matrixfile")
        p.add_argument("--k", action="store", required=True, type=int, metavar="k")
        p.add_argument("--dictfile", action="store", required=False, metavar="dictfile", default="")
        p.add_argument("--hiernmf2", action="store", required=False, metavar="hiernmf2", default=0, choices=[0, 1])
        p.add_argument("--algorithm", action="store", required=False, default="BPP", metavar="algorithm",
                       choices=["MU", "HALS", "RANK2", "BPP"])
        p.add_argument("--stopping", action="store", required=False, metavar="stopping", default="PG_RATIO",
                       choices=["PG_RATIO", "DELTA"])
        p.add_argument("--tol", action="store", type=float, required=False, metavar="tol", default=0.005)
        p.add_argument("--tolcount", action="store", type=int, required=False, metavar="tolcount", default=1)
        p.add_argument("--infile_W", action="store", required=False, metavar="infile_W", default="")
        p.add_argument("--infile_H", action="store", required=False, metavar="infile_H", default="")
        p.add_argument("--outfile_W", action="store", required=False, metavar="outfile_W", default="w.csv")
        p.add_argument("--outfile_H", action="store", required=False, metavar="outfile_H", default="h.csv")
        p.add_argument("--outprecision", action="store", type=int, required=False, metavar="outprecision", default=6)
        p.add_argument("--maxiter", action="store", type=int, required=False, metavar="maxiter", default=5000)
        p.add_argument("--miniter", action="store", type=int, required=False, metavar="miniter", default=5)
        p.add_argument("--maxthreads", action="store", type=int, required=False, metavar="maxthreads", default=8)
        p.add_argument("--maxterms", action="store", type=int, required=False, metavar="maxterms", default=5)
        p.add_argument("--normalize", action="store", type=int, required=False, metavar="normalize", default=1)
        p.add_argument("--verbose", action="store", type=int, required=False, metavar="verbose", default=1)
        return p.parse_args()

    # -- versions ---------------------------------------------------------------------------
    def get_major_version(self):
        return L.lib().smk_api_get_major_version()

    def get_minor_version(self):
        return L.lib().smk_api_get_minor_version()

    def get_patch_level(self):
        return L.lib().smk_api_get_patch_level()

    def get_version_string(self):
        return f"{self.get_major_version()}.{self.get_minor_version()}.{self.get_patch_level()}"

    # -- input ------------------------------------------------------------------------------
    def load_matrix(self, filepath="", height=0, width=0, delim="", buffer=[], matrix=[], nz=0,
                    row_indices=[], col_offsets=[], column_major=False, sparse_matrix=None):
        if len(row_indices) > 0 and len(col_offsets) > 0:
            self._load_sparse_buffer(height, width, nz, buffer, row_indices, col_offsets)
        elif len(buffer) > 0 and height != 0 and width != 0:
            self._load_dense_buffer(buffer, height, width)
        elif len(matrix) > 0:
            self._load_numpy(matrix, column_major)
        if filepath != "":
            self._load_matrix_file(filepath)

    def nmf(self, k, algorithm, infile_W="", infile_H="", precision=4, min_iter=5, max_iter=5000, tol=0.005,
            max_threads=8, outdir="."):
        if self.is_matrix_loaded():
            l = L.lib()
            l.smk_api_set_output_precision(precision)
            l.smk_api_set_min_iter(min_iter)
            l.smk_api_set_max_iter(max_iter)
            _raise_if(l.smk_api_set_nmf_tolerance(tol), "SetNmfTolerance")
            l.smk_api_set_max_threads(max_threads)
            _raise_if(l.smk_api_set_output_dir(_b(outdir)), "SetOutputDir")
            _raise_if(l.smk_api_nmf(int(k), _ALG[algorithm.upper()], _b(infile_W), _b(infile_H)), "Nmf")
        else:
            print("Error: No matrix loaded, not running NMF.")

    def get_inputs(self):
        l = L.lib()
        return {
            "precision": l.smk_api_get_output_precision(),
            "min_iter": l.smk_api_get_min_iter(),
            "max_iter": l.smk_api_get_max_iter(),
            "tol": l.smk_api_get_nmf_tolerance(),
            "max_threads": l.smk_api_get_max_threads(),
            "outdir": l.smk_api_get_output_dir().decode(),
            "format": "XML" if l.smk_api_get_output_format() == 0 else "JSON",
        }

    def is_matrix_loaded(self):
        return bool(L.lib().smk_api_is_matrix_loaded())

    def finalize(self):
        L.lib().smk_api_finalize()

    def _locked(self, fn):
        ld, h, w = C.c_uint(0), C.c_uint(0), C.c_uint(0)
        ptr = fn(C.byref(ld), C.byref(h), C.byref(w))
        if not ptr or h.value == 0 or w.value == 0:
            return np.zeros((0, 0))
        flat = np.ctypeslib.as_array(ptr, shape=(h.value * w.value,))
        return np.array(flat).reshape(w.value, h.value).T     # column-major buffer -> (height, width)

    def get_H(self):
        return self._locked(L.lib().smk_api_locked_buffer_h)

    def get_W(self):
        return self._locked(L.lib().smk_api_locked_buffer_w)

    def hiernmf2(self, k, format="XML", maxterms=5, tol=0.0001):
        if self._dictionary_loaded and self.is_matrix_loaded():
            l = L.lib()
            _raise_if(l.smk_api_set_hiernmf2_tolerance(tol), "SetHierNmf2Tolerance")
            l.smk_api_set_max_terms(maxterms)
            l.smk_api_set_output_format(0 if format.lower() == "xml" else 1)
            _raise_if(l.smk_api_hiernmf2(int(k)), "HierNmf2")
        else:
            print("Error: No dictionary loaded.")

    def load_dictionary(self, filepath="", dictionary=[]):
        if filepath != "":
            _raise_if(L.lib().smk_api_load_dictionary_file(_b(filepath)), "LoadDictionary")
            self._dictionary_loaded = True
        elif len(dictionary) > 0:
            terms = (C.c_char_p * len(dictionary))(*[_b(str(t)) for t in dictionary])
            _raise_if(L.lib().smk_api_load_dictionary(terms, len(dictionary)), "LoadDictionary")
            self._dictionary_loaded = True
        else:
            print("Error: Invalid dictionary arguments.")

    # -- MI355X extensions --------------------------------------------------------------------
    def set_device_storage(self, storage="f32"):
        L.lib().smk_api_set_device_storage(1 if str(storage).lower() == "bf16" else 0)

    def get_iteration_count(self):
        return L.lib().smk_api_get_iteration_count()

    def seed_rng(self, seed):
        L.lib().smk_api_seed_rng(int(seed))

    # -- private ------------------------------------------------------------------------------
    def _load_matrix_file(self, filepath):
        _raise_if(L.lib().smk_api_load_matrix_file(_b(filepath)), "LoadMatrix")

    def _load_dense_buffer(self, buffer, height, width):
        buf = np.ascontiguousarray(buffer, dtype=np.float64)
        _raise_if(L.lib().smk_api_load_matrix_dense(buf.ctypes.data_as(C.POINTER(C.c_double)), height, height, width),
                  "LoadMatrix")

    def _load_sparse_buffer(self, height, width, nz, buffer, row_indices, col_offsets):
        d = np.ascontiguousarray(buffer, dtype=np.float64)
        ri = np.ascontiguousarray(row_indices, dtype=np.uint32)
        co = np.ascontiguousarray(col_offsets, dtype=np.uint32)
        _raise_if(L.lib().smk_api_load_matrix_sparse(height, width, nz, d.ctypes.data_as(C.POINTER(C.c_double)),
                                                     ri.ctypes.data_as(C.POINTER(C.c_uint)),
                                                     co.ctypes.data_as(C.POINTER(C.c_uint))), "LoadMatrix")

    def _load_numpy(self, matrix, column_major):
        # The reference hands the transposed array's shape to LoadMatrix (smallk_lib.pyx:864-868),
        # which is only right for square input (and untested there).  The documented intent is
        # "load this height x width matrix": do that.
        m = np.asarray(matrix, dtype=np.float64)
        col = np.asfortranarray(m)
        _raise_if(L.lib().smk_api_load_matrix_dense(col.ctypes.data_as(C.POINTER(C.c_double)), col.shape[0],
                                                    col.shape[0], col.shape[1]), "LoadMatrix")
