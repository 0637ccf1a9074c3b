"""Object wrappers over the C ABI: device-resident A and the NmfSolve<> solver.

Host-side mirror of the reference's inner seam (common/include/nmf.hpp:77-81,
common/include/nmf_solve_generic.hpp:34-140).  Numpy in/out, fp64, column-major.
All compute happens in libsmallk_amd.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L

ALGORITHMS = {"MU": L.ALG_MU, "HALS": L.ALG_HALS, "RANK2": L.ALG_RANK2, "BPP": L.ALG_BPP}
STORAGE = {"f32": L.STORE_F32, "fp32": L.STORE_F32, "bf16": L.STORE_BF16}


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def initialize(device: int = -1):
    """NmfInitialize (nmf.hpp:71): select the GPU, create the stream.  Raises without a GPU."""
    L.check(L.lib().smk_initialize(device), "smk_initialize")


def is_initialized() -> bool:
    return L.lib().smk_is_initialized() == L.INITIALIZED


def finalize():
    L.lib().smk_finalize()


def thread_context_begin(device: int = -1):
    """this host thread gets a device context of its own (stream, handles) until thread_context_end()"""
    L.check(L.lib().smk_thread_context_begin(device), "smk_thread_context_begin")


def thread_context_end():
    L.lib().smk_thread_context_end()


def trim_device_cache() -> int:
    """Return the library's cached (freed) device workspaces of the current device to the HIP runtime; bytes that were cached."""
    return int(L.lib().smk_device_trim())


def set_stream(stream_ptr: int):
    L.check(L.lib().smk_set_stream(C.c_void_p(stream_ptr)), "smk_set_stream")


def uniform_host(rows, cols, seed, *, quant=0, r0=0, c0=0, gheight=None) -> np.ndarray:
    """Counter-based uniform [0,1) matrix on the host (RandomMatrix stand-in)."""
    out = np.empty((rows, cols), order="F")
    L.lib().smk_uniform_fill_host(_p(out), rows, rows, cols, r0, c0, rows if gheight is None else gheight, seed, quant)
    return out


def make_options(m, n, k, algorithm, *, min_iter=5, max_iter=5000, tol=0.005, tolcount=1,
                 prog_est=None, normalize=True, max_threads=1, verbose=False) -> L.Options:
    alg = ALGORITHMS[algorithm] if isinstance(algorithm, str) else int(algorithm)
    if prog_est is None:   # smallk::Nmf's rule (smallk/src/smallk.cpp:581-584)
        prog_est = L.PROG_DELTA_FNORM if alg == L.ALG_MU else L.PROG_PG_RATIO
    return L.Options(tol, alg, prog_est, m, n, k, min_iter, max_iter, tolcount, max_threads,
                     int(verbose), int(normalize))


class DenseMatrix:
    """A (or the column shard [col0, col0+ncols) of it) resident in HBM with its transpose."""

    def __init__(self, height, width_global, *, col0=0, ncols=None, storage="f32", single_copy=False):
        """single_copy: no stored transpose (bf16 or fp32 storage; serves MU, HALS and BPP with the 16-bit product forms; RANK2 and the
        accurate form build the transpose on demand -- smk_matrix_create_single_copy)"""
        self.height = int(height)
        self.width_global = int(width_global)
        self.col0 = int(col0)
        self.ncols = int(width_global - col0 if ncols is None else ncols)
        self.storage = STORAGE[storage] if isinstance(storage, str) else int(storage)
        self._h = C.c_void_p()
        create = L.lib().smk_matrix_create_single_copy if single_copy else L.lib().smk_matrix_create
        L.check(create(C.byref(self._h), self.height, self.width_global, self.col0, self.ncols, self.storage),
                "smk_matrix_create_single_copy" if single_copy else "smk_matrix_create")

    @property
    def single_copy(self) -> bool:
        return bool(L.lib().smk_matrix_is_single_copy(self._h))

    @property
    def device_bytes(self) -> int:
        return int(L.lib().smk_matrix_device_bytes(self._h))

    @classmethod
    def from_host(cls, A, *, storage="f32", single_copy=False):
        A = _f(A)
        mat = cls(A.shape[0], A.shape[1], storage=storage, single_copy=single_copy)
        mat.upload(A)
        return mat

    def upload(self, A_local):
        A_local = _f(A_local)
        assert A_local.shape == (self.height, self.ncols)
        L.check(L.lib().smk_matrix_upload_f64(self._h, _p(A_local), A_local.shape[0]), "smk_matrix_upload_f64")

    def fill_uniform(self, seed):
        L.check(L.lib().smk_matrix_fill_uniform(self._h, seed), "smk_matrix_fill_uniform")

    def fill_planted(self, seed, kstar, threshold=0.7, noise=0.05):
        """A = Ws Hs + noise U with sparse planted factors (entries <= threshold dropped), SURVEY 8(d)."""
        L.check(L.lib().smk_matrix_fill_planted(self._h, seed, kstar, threshold, noise), "smk_matrix_fill_planted")

    def download(self) -> np.ndarray:
        out = np.empty((self.height, self.ncols), order="F")
        L.check(L.lib().smk_matrix_download_f64(self._h, _p(out), self.height), "smk_matrix_download_f64")
        return out

    def close(self):
        if self._h:
            L.lib().smk_matrix_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SparseMatrix(DenseMatrix):
    """Sparse A in CSC (scipy.sparse.csc_matrix or (data, indices, indptr, shape)), resident in HBM with
    its transpose.  Shares the NmfSolver interface with DenseMatrix."""

    def __init__(self, data, indices, indptr, shape, *, col0=0, width_global=None):
        self.height = int(shape[0])
        self.ncols = int(shape[1])
        self.width_global = int(width_global if width_global is not None else shape[1])
        self.col0 = int(col0)
        self.storage = L.STORE_F32
        d = np.ascontiguousarray(data, dtype=np.float64)
        ri = np.ascontiguousarray(indices, dtype=np.uint32)
        co = np.ascontiguousarray(indptr, dtype=np.uint32)
        self.nnz = int(co[-1] - co[0])
        self._h = C.c_void_p()
        L.check(L.lib().smk_matrix_create_sparse(C.byref(self._h), self.height, self.width_global, self.col0,
                                                 self.ncols, self.nnz, co.ctypes.data_as(C.POINTER(C.c_uint)),
                                                 ri.ctypes.data_as(C.POINTER(C.c_uint)), _p(d)),
                "smk_matrix_create_sparse")

    @classmethod
    def from_scipy(cls, A):
        A = A.tocsc()
        return cls(A.data, A.indices, A.indptr, A.shape)

    def product(self, X, *, transposed=False, reps=0):
        """The sparse Gemm by itself: X (k x height) * A, or with ``transposed`` X (k x width) * A'; returns the k x ncols
        result and, for reps > 0, the average launch time in ms (smk_matrix_sparse_product)."""
        X = _f(X)
        k = X.shape[0]
        assert X.shape[1] == (self.ncols if transposed else self.height)
        out = np.zeros((k, self.height if transposed else self.ncols), order="F")
        ms = C.c_double(0)
        L.check(L.lib().smk_matrix_sparse_product(self._h, int(transposed), k, _p(X), k, _p(out), k, int(reps),
                                                  C.byref(ms) if reps > 0 else None), "smk_matrix_sparse_product")
        return (out, ms.value) if reps > 0 else out


def load_matrix_market(path):
    """MatrixMarket coordinate file -> (data, indices, indptr, (height, width)) in CSC."""
    h, w, nz = C.c_uint(0), C.c_uint(0), C.c_uint(0)
    p = str(path).encode()
    if L.lib().smk_load_matrix_market(p, C.byref(h), C.byref(w), C.byref(nz), None, None, None) != 1:
        raise RuntimeError(f"could not load MatrixMarket file {path}")
    indptr = np.zeros(w.value + 1, dtype=np.uint32)
    indices = np.zeros(nz.value, dtype=np.uint32)
    data = np.zeros(nz.value, dtype=np.float64)
    L.lib().smk_load_matrix_market(p, C.byref(h), C.byref(w), C.byref(nz), indptr.ctypes.data_as(C.POINTER(C.c_uint)),
                                   indices.ctypes.data_as(C.POINTER(C.c_uint)), _p(data))
    return data, indices, indptr, (h.value, w.value)


def nmf_sparse(A, W0, H0, algorithm, **kw):
    """One-shot sparse NMF = ``NmfSparse(...)`` (common/src/nmf.cpp:232-300); A is a scipy sparse matrix."""
    A = A.tocsc()
    W = _f(W0).copy(order="F")
    H = _f(H0).copy(order="F")
    m, n = A.shape
    k = W.shape[1]
    o = make_options(m, n, k, algorithm, **kw)
    st = L.Stats()
    d = np.ascontiguousarray(A.data, dtype=np.float64)
    ri = np.ascontiguousarray(A.indices, dtype=np.uint32)
    co = np.ascontiguousarray(A.indptr, dtype=np.uint32)
    rc = L.lib().smk_nmf_sparse(C.byref(o), m, n, d.size, co.ctypes.data_as(C.POINTER(C.c_uint)),
                                ri.ctypes.data_as(C.POINTER(C.c_uint)), _p(d), _p(W), m, _p(H), k, C.byref(st))
    if rc not in (L.OK, L.FAILURE, L.BAD_PARAM, L.NOTINITIALIZED, L.SIZE_TOO_LARGE):
        L.check(rc, "smk_nmf_sparse")
    return NmfResult(rc, W, H, st.iteration_count, st.elapsed_us)


@dataclass
class NmfResult:
    result: int
    W: np.ndarray
    H: np.ndarray
    iteration_count: int
    elapsed_us: int


class NmfSolver:
    """One NmfSolve<> instance bound to a DenseMatrix."""

    def __init__(self, A: DenseMatrix, options: L.Options):
        self.A = A
        self.options = options
        self.k = options.k
        self._h = C.c_void_p()
        self._cb = None
        L.check(L.lib().smk_solver_create(C.byref(self._h), C.byref(options), A._h), "smk_solver_create")

    def set_factors(self, W0, H0_local):
        W0, H0 = _f(W0), _f(H0_local)
        assert W0.shape == (self.A.height, self.k) and H0.shape == (self.k, self.A.ncols)
        L.check(L.lib().smk_solver_set_factors(self._h, _p(W0), W0.shape[0], _p(H0), H0.shape[0]),
                "smk_solver_set_factors")

    def set_factors_uniform(self, seed_w, seed_h):
        """W0 / H0 = uniform_host(m, k, seed_w) / uniform_host(k, n, seed_h), generated on the device"""
        L.check(L.lib().smk_solver_set_factors_uniform(self._h, seed_w, seed_h), "smk_solver_set_factors_uniform")

    def run(self):
        st = L.Stats()
        rc = L.lib().smk_solver_run(self._h, C.byref(st))
        return rc, st.iteration_count, st.elapsed_us

    def iterate(self, iters):
        L.check(L.lib().smk_solver_iterate(self._h, iters), "smk_solver_iterate")

    def iterate_checked(self, iters) -> float:
        """`iters` iterations with the stopping rule's metric formed and read back after each (never stopping); returns the last metric"""
        v = C.c_double(0)
        L.check(L.lib().smk_solver_iterate_checked(self._h, iters, C.byref(v)), "smk_solver_iterate_checked")
        return v.value

    def sync(self):
        return L.lib().smk_solver_sync(self._h)

    def progress(self) -> float:
        v = C.c_double(0)
        L.check(L.lib().smk_solver_progress(self._h, C.byref(v)), "smk_solver_progress")
        return v.value

    def factors(self, normalize=False):
        W = np.empty((self.A.height, self.k), order="F")
        H = np.empty((self.k, self.A.ncols), order="F")
        rc = L.lib().smk_solver_get_factors(self._h, int(normalize), _p(W), W.shape[0], _p(H), H.shape[0])
        if rc not in (L.OK, L.FAILURE):
            L.check(rc, "smk_solver_get_factors")
        return W, H

    def product_form(self):
        """(form, guard_checks, guard_fired, cond x delta of the last check); form: 3 bf16x3, 4 fp16 two-term, 8 accurate"""
        c, f, v = C.c_int(0), C.c_int(0), C.c_double(0)
        form = L.lib().smk_solver_product_form(self._h, C.byref(c), C.byref(f), C.byref(v))
        return form, c.value, f.value, v.value

    def enable_timing(self, on=True):
        L.check(L.lib().smk_solver_enable_timing(self._h, int(on)), "smk_solver_enable_timing")

    def kernel_time(self, which):
        ms, cnt = C.c_double(0), C.c_int(0)
        L.check(L.lib().smk_solver_kernel_time(self._h, which, C.byref(ms), C.byref(cnt)), "smk_solver_kernel_time")
        return ms.value, cnt.value

    def kernel_name(self, which) -> str:
        """the kernel pass `which` (0 = W'A, 1 = H*At) launches; which = 2: how the stopping-rule checks were formed so far (counts per route)"""
        buf = C.create_string_buffer(160)
        L.check(L.lib().smk_solver_kernel_name(self._h, which, buf, 160), "smk_solver_kernel_name")
        return buf.value.decode()

    def kernel_work(self, which):
        b, f = C.c_double(0), C.c_double(0)
        L.check(L.lib().smk_solver_kernel_work(self._h, which, C.byref(b), C.byref(f)), "smk_solver_kernel_work")
        return b.value, f.value

    def comm_workspace_bytes(self) -> int:
        n = C.c_size_t(0)
        L.check(L.lib().smk_solver_comm_workspace_bytes(self._h, C.byref(n)), "smk_solver_comm_workspace_bytes")
        return n.value

    def set_comm(self, rank, world, callback, workspace_ptr, workspace_bytes):
        """callback(ptr:int, count:int, dtype:int) -> 0 on success; kept alive by this object."""
        def _tramp(_user, ptr, count, dtype):
            try:
                return int(callback(ptr, count, dtype) or 0)
            except Exception as e:  # never let an exception cross the C boundary
                print("all-reduce callback failed:", e, flush=True)
                return 1
        self._cb = L.ALLREDUCE_FN(_tramp)
        L.check(L.lib().smk_solver_set_comm(self._h, rank, world, self._cb, None, C.c_void_p(workspace_ptr),
                                            workspace_bytes), "smk_solver_set_comm")

    def attach_comm(self, comm):
        """Native collectives (RCCL / in-process stand-in); call before set_factors()."""
        L.check(L.lib().smk_solver_attach_comm(self._h, comm._h), "smk_solver_attach_comm")
        self._comm = comm          # keep alive

    def close(self):
        if self._h:
            L.lib().smk_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def nmf(A, W0, H0, algorithm, *, storage="f32", **kw) -> NmfResult:
    """One-shot dense NMF = ``Nmf(NmfOptions, A, W, H, stats)`` (common/src/nmf.cpp:173-229)."""
    A = _f(A)
    W = _f(W0).copy(order="F")
    H = _f(H0).copy(order="F")
    m, n = A.shape
    k = W.shape[1]
    o = make_options(m, n, k, algorithm, **kw)
    st = L.Stats()
    stg = STORAGE[storage] if isinstance(storage, str) else int(storage)
    rc = L.lib().smk_nmf_dense(C.byref(o), _p(A), m, _p(W), m, _p(H), k, C.byref(st), stg)
    if rc not in (L.OK, L.FAILURE, L.BAD_PARAM, L.NOTINITIALIZED, L.SIZE_TOO_LARGE):
        L.check(rc, "smk_nmf_dense")
    return NmfResult(rc, W, H, st.iteration_count, st.elapsed_us)


def nnls_blockpivot(LHS, RHS, Xinit):
    """``NnlsBlockpivot`` (common/include/nnls.hpp:144-244) on the device, by itself.

    LHS k x k SPD, RHS k x ncols, Xinit the warm start (passive set = Xinit > 0).
    Returns (ok, X, Y) with Y = LHS X - RHS; ok False = the reference's ``false``."""
    G = _f(LHS)
    B = _f(RHS)
    X = _f(Xinit).copy(order="F")
    k, ncols = B.shape
    Y = np.zeros((k, ncols), order="F")
    rc = L.lib().smk_nnls_blockpivot(k, ncols, _p(G), k, _p(B), k, _p(X), k, _p(Y), k)
    if rc not in (L.OK, L.FAILURE):
        L.check(rc, "smk_nnls_blockpivot")
    return rc == L.OK, X, Y


class Comm:
    """A communicator of the column-sharded solver (include/smallk_amd.h, multi-GPU section): RCCL, or the
    in-process stand-in for several shards on one device."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def unique_id(cls) -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.lib().smk_comm_unique_id(buf), "smk_comm_unique_id")
        return buf.raw

    @classmethod
    def init_rank(cls, uid: bytes, rank: int, world: int):
        h = C.c_void_p()
        L.check(L.lib().smk_comm_init_rank(C.byref(h), C.create_string_buffer(uid, 128), rank, world), "smk_comm_init_rank")
        return cls(h)

    @classmethod
    def init_all(cls, ndev: int, devices=None):
        hs = (C.c_void_p * ndev)()
        dv = (C.c_int * ndev)(*devices) if devices is not None else None
        L.check(L.lib().smk_comm_init_all(hs, ndev, dv), "smk_comm_init_all")
        return [cls(C.c_void_p(h)) for h in hs]

    @classmethod
    def init_local(cls, nranks: int):
        hs = (C.c_void_p * nranks)()
        L.check(L.lib().smk_comm_init_local(hs, nranks), "smk_comm_init_local")
        return [cls(C.c_void_p(h)) for h in hs]

    def selftest(self):
        """known values through one all-reduce and one all-gather; every rank calls it; raises on a wrong answer"""
        L.check(L.lib().smk_comm_selftest(self._h), "smk_comm_selftest")

    @property
    def rank(self):
        return L.lib().smk_comm_rank(self._h)

    @property
    def world(self):
        return L.lib().smk_comm_world(self._h)

    def close(self):
        if self._h:
            L.lib().smk_comm_destroy(self._h)
            self._h = None


def nmf_sharded(A, W0, H0, algorithm, nshards, *, storage="f32", devices=None, local_stub=False, **kw) -> NmfResult:
    """``Nmf(...)`` on ``nshards`` column shards, one host thread and one device per shard (RCCL), or all shards on
    the current device through the in-process stand-in (``local_stub=True``)."""
    A = _f(A)
    W = _f(W0).copy(order="F")
    H = _f(H0).copy(order="F")
    m, n = A.shape
    k = W.shape[1]
    o = make_options(m, n, k, algorithm, **kw)
    st = L.Stats()
    stg = STORAGE[storage] if isinstance(storage, str) else int(storage)
    dv = (C.c_int * nshards)(*devices) if devices is not None else None
    rc = L.lib().smk_nmf_dense_sharded(C.byref(o), _p(A), m, _p(W), m, _p(H), k, C.byref(st), stg, nshards, dv,
                                       1 if local_stub else 0)
    if rc not in (L.OK, L.FAILURE, L.BAD_PARAM, L.NOTINITIALIZED, L.SIZE_TOO_LARGE):
        L.check(rc, "smk_nmf_dense_sharded")
    return NmfResult(rc, W, H, st.iteration_count, st.elapsed_us)
