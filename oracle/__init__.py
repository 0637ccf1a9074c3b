"""oracle -- CPU checker for the dense NMF hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``smallk_amd``) never does.

PARITY UNPINNED: the reference ships no golden vectors for this path and cannot
be compiled in this image (Elemental missing), see ``oracle/nmf_oracle.c`` header
and DESIGN.md section 3.

The heavy lifting is the plain-C restatement in ``nmf_oracle.c`` (built by
``make -C oracle`` into ``oracle/_build/liboracle.so``); this module is a thin
ctypes veneer with numpy in/out.  All matrices are float64, column-major
(Fortran order), exactly like the reference's buffers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

# enums: common/include/nmf.hpp:17-41
OK, NOTINITIALIZED, INITIALIZED, BAD_PARAM, FAILURE, SIZE_TOO_LARGE = 0, -1, -2, -3, -4, -5
MU, HALS, RANK2, BPP = 0, 1, 2, 3
PG_RATIO, DELTA_FNORM = 0, 1
ALGORITHMS = {"MU": MU, "HALS": HALS, "RANK2": RANK2, "BPP": BPP}


class _Options(C.Structure):
    _fields_ = [("tol", C.c_double), ("algorithm", C.c_int), ("prog_est_algorithm", C.c_int),
                ("height", C.c_int), ("width", C.c_int), ("k", C.c_int),
                ("min_iter", C.c_int), ("max_iter", C.c_int), ("tolcount", C.c_int),
                ("max_threads", C.c_int), ("verbose", C.c_int), ("normalize", C.c_int)]


class _Stats(C.Structure):
    _fields_ = [("elapsed_us", C.c_ulonglong), ("iteration_count", C.c_int)]


def build(force: bool = False) -> str:
    """Compile the C restatement (and, when /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "nmf_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    if os.path.isdir("/root/reference/common/src"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        i64 = C.c_int64
        _lib.orc_nmf.restype = C.c_int
        _lib.orc_nmf.argtypes = [C.POINTER(_Options), dp, i64, dp, i64, dp, i64, C.POINTER(_Stats), dp]
        up = C.POINTER(C.c_uint)
        _lib.orc_nmf_sparse.restype = C.c_int
        _lib.orc_nmf_sparse.argtypes = [C.POINTER(_Options), up, up, dp, dp, i64, dp, i64, C.POINTER(_Stats), dp]
        _lib.orc_nnls_blockpivot.restype = C.c_int
        _lib.orc_nnls_blockpivot.argtypes = [C.c_int, i64, dp, C.c_int, dp, i64, dp, i64, dp, i64, C.POINTER(C.c_int)]
        _lib.orc_fill_uniform.restype = None
        _lib.orc_fill_uniform.argtypes = [dp, i64, i64, i64, i64, i64, i64, C.c_uint64, C.c_int]
        _lib.orc_fill_planted.restype = None
        _lib.orc_fill_planted.argtypes = [dp, i64, i64, i64, i64, i64, i64, C.c_uint64, C.c_int, C.c_double, C.c_double, C.c_int]
        _lib.orc_quantize.restype = None
        _lib.orc_quantize.argtypes = [dp, i64, C.c_int]
        _lib.orc_projected_gradient_norm.restype = C.c_double
        _lib.orc_projected_gradient_norm.argtypes = [i64, i64, C.c_int, dp, i64, dp, i64, dp, i64, dp, i64]
        _lib.orc_normalize_and_scale.restype = C.c_int
        _lib.orc_normalize_and_scale.argtypes = [i64, i64, C.c_int, dp, i64, dp, i64]
        _lib.orc_gemm.restype = None
        _lib.orc_gemm.argtypes = [C.c_int, C.c_int, i64, i64, i64, C.c_double, dp, i64, dp, i64, C.c_double, dp, i64]
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_big_product_time.restype = None
        _lib.orc_big_product_time.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int)]
        _lib.orc_set_num_threads.restype = None
        _lib.orc_set_num_threads.argtypes = [C.c_int]
        # hosts with hundreds of cores: an OpenMP team per tiny loop is pathological
        _lib.orc_set_num_threads(int(os.environ.get("ORACLE_THREADS", min(os.cpu_count() or 1, 16))))
    return _lib


def _f(a) -> np.ndarray:
    return np.asfortranarray(a, dtype=np.float64)


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@dataclass
class NmfResult:
    result: int
    W: np.ndarray
    H: np.ndarray
    iteration_count: int
    elapsed_us: int
    metrics: np.ndarray


def nmf(A, W0, H0, algorithm, *, min_iter=5, max_iter=5000, tol=0.005, tolcount=1,
        prog_est=None, normalize=True, max_threads=0, verbose=False) -> NmfResult:
    """Restatement of ``Nmf(NmfOptions, A, W, H, stats)`` (common/src/nmf.cpp:173-229)."""
    alg = ALGORITHMS[algorithm] if isinstance(algorithm, str) else int(algorithm)
    if prog_est is None:   # smallk::Nmf's rule, smallk/src/smallk.cpp:581-584
        prog_est = DELTA_FNORM if alg == MU else PG_RATIO
    A = _f(A)
    W = _f(W0).copy(order="F")
    H = _f(H0).copy(order="F")
    m, n = A.shape
    k = W.shape[1]
    o = _Options(tol, alg, prog_est, m, n, k, min_iter, max_iter, tolcount, max_threads,
                 int(verbose), int(normalize))
    st = _Stats()
    metrics = np.full(max(max_iter, 1), np.nan)
    rc = lib().orc_nmf(C.byref(o), _p(A), A.shape[0], _p(W), W.shape[0], _p(H), H.shape[0],
                       C.byref(st), _p(metrics))
    return NmfResult(rc, W, H, st.iteration_count, st.elapsed_us, metrics)


def nmf_sparse(A, W0, H0, algorithm, *, min_iter=5, max_iter=5000, tol=0.005, tolcount=1,
               prog_est=None, normalize=True, max_threads=0, verbose=False) -> NmfResult:
    """Restatement of ``NmfSparse`` (common/src/nmf.cpp:232-300): A is a scipy CSC matrix; the driver and the
    solvers are the dense ones, only the three products with A run over the stored entries."""
    alg = ALGORITHMS[algorithm] if isinstance(algorithm, str) else int(algorithm)
    if prog_est is None:
        prog_est = DELTA_FNORM if alg == MU else PG_RATIO
    A = A.tocsc()
    cp = np.ascontiguousarray(A.indptr, dtype=np.uint32)
    ri = np.ascontiguousarray(A.indices, dtype=np.uint32)
    va = np.ascontiguousarray(A.data, dtype=np.float64)
    if len(ri) == 0:
        ri, va = np.zeros(1, np.uint32), np.zeros(1)
    W = _f(W0).copy(order="F")
    H = _f(H0).copy(order="F")
    m, n = A.shape
    k = W.shape[1]
    o = _Options(tol, alg, prog_est, m, n, k, min_iter, max_iter, tolcount, max_threads, int(verbose), int(normalize))
    st = _Stats()
    metrics = np.full(max(max_iter, 1), np.nan)
    up = C.POINTER(C.c_uint)
    rc = lib().orc_nmf_sparse(C.byref(o), cp.ctypes.data_as(up), ri.ctypes.data_as(up), _p(va), _p(W), W.shape[0],
                              _p(H), H.shape[0], C.byref(st), _p(metrics))
    return NmfResult(rc, W, H, st.iteration_count, st.elapsed_us, metrics)


def nnls_blockpivot(LHS, RHS, Xinit):
    """Restatement of ``NnlsBlockpivot`` (nnls.hpp:144-244).  Returns (ok, X, Y, pivots)."""
    LHS = _f(LHS)
    RHS = _f(RHS)
    X = _f(Xinit).copy(order="F")
    k, ncols = RHS.shape
    Y = np.zeros((k, ncols), order="F")
    piv = C.c_int(0)
    ok = lib().orc_nnls_blockpivot(k, ncols, _p(LHS), k, _p(RHS), k, _p(X), k, _p(Y), k, C.byref(piv))
    return bool(ok), X, Y, piv.value


def fill_uniform(rows, cols, seed, *, quant=0, r0=0, c0=0, gheight=None) -> np.ndarray:
    """Counter-based uniform [0,1) block; bit-identical to the device generator."""
    out = np.empty((rows, cols), order="F")
    lib().orc_fill_uniform(_p(out), rows, rows, cols, r0, c0, rows if gheight is None else gheight,
                           seed, quant)
    return out


def fill_planted(rows, cols, seed, kstar, *, threshold=0.7, noise=0.05, quant=0, r0=0, c0=0, gheight=None) -> np.ndarray:
    """Planted low-rank + noise block (orc_fill_planted); bit-identical to smk_matrix_fill_planted.  Rows r0.., columns c0..
    of the matrix of global height `gheight`."""
    out = np.empty((rows, cols), order="F")
    lib().orc_fill_planted(_p(out), rows, rows, cols, r0, c0, rows if gheight is None else gheight, seed, kstar,
                           threshold, noise, quant)
    return out


def quantize(a, quant) -> np.ndarray:
    """Round to what the device stores: quant 0 -> fp32, 1 -> bf16 (RNE); returned as float64."""
    out = _f(a).copy(order="F")
    lib().orc_quantize(_p(out), out.size, quant)
    return out


def projected_gradient_norm(gradW, gradH, W, H) -> float:
    gradW, gradH, W, H = _f(gradW), _f(gradH), _f(W), _f(H)
    m, k = W.shape
    n = H.shape[1]
    return lib().orc_projected_gradient_norm(m, n, k, _p(gradW), m, _p(gradH), k, _p(W), m, _p(H), k)


def gemm_tn(X, A) -> np.ndarray:
    """X' A with the oracle's own GEMM loops (timing of the plain port in bench.py's cpu_baseline)."""
    X, A = _f(X), _f(A)
    m, k = X.shape
    n = A.shape[1]
    out = np.zeros((k, n), order="F")
    lib().orc_gemm(1, 0, k, n, m, 1.0, _p(X), m, _p(A), m, 0.0, _p(out), k)
    return out


def gemm_nt(A, H) -> np.ndarray:
    """A H' with the oracle's own GEMM loops."""
    A, H = _f(A), _f(H)
    m, n = A.shape
    k = H.shape[0]
    out = np.zeros((m, k), order="F")
    lib().orc_gemm(0, 1, m, k, n, 1.0, _p(A), m, _p(H), k, 0.0, _p(out), m)
    return out


def big_product_time():
    """(seconds, calls) spent in the products with A during the last nmf() / nmf_sparse() call."""
    t, c = C.c_double(0), C.c_int(0)
    lib().orc_big_product_time(C.byref(t), C.byref(c))
    return t.value, c.value


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))
