"""smallk_amd -- MI355X-native dense NMF (MU / HALS / BPP) behind the libsmallk API.

The numeric path is hand-written HIP for gfx950 in ``smallk_amd/csrc`` behind the C ABI
``include/smallk_amd.h``; this package is the host-side mirror of the reference's Python
surface (pysmallk ``SmallkAPI``) plus thin object wrappers.  No CPU fallback exists.
"""
from . import _lib
from .solver import (DenseMatrix, SparseMatrix, NmfSolver, NmfResult, nmf, nmf_sparse, load_matrix_market,
                     initialize, finalize, is_initialized, make_options, uniform_host, set_stream,
                     nnls_blockpivot, nmf_sharded, Comm, thread_context_begin, thread_context_end)
from .api import SmallkAPI
from . import hierclust
from . import flatclust
from .hierclust import hier_nmf2, TreeResults

__all__ = ["DenseMatrix", "SparseMatrix", "nmf_sparse", "load_matrix_market", "NmfSolver", "NmfResult", "nmf", "initialize", "finalize", "is_initialized",
           "make_options", "uniform_host", "nnls_blockpivot", "nmf_sharded", "Comm", "set_stream", "thread_context_begin", "thread_context_end", "SmallkAPI", "hierclust", "flatclust", "hier_nmf2", "TreeResults", "_lib"]
