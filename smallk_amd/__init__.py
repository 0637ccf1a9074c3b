"""smallk_amd -- MI355X-native dense NMF (MU / HALS / BPP) behind the libsmallk API.

The numeric path is hand-written HIP for gfx950 in ``smallk_amd/csrc`` behind the C ABI
``include/smallk_amd.h``; this package is the host-side mirror of the reference's Python
surface (pysmallk ``SmallkAPI``) plus thin object wrappers.  No CPU fallback exists.
"""
from . import _lib
from .solver import (DenseMatrix, SparseMatrix, NmfSolver, NmfResult, nmf, nmf_sparse, load_matrix_market,
                     initialize, finalize, is_initialized, make_options, uniform_host, set_stream,
                     nnls_blockpivot, nmf_sharded, Comm, thread_context_begin, thread_context_end, trim_device_cache)
from .api import SmallkAPI
from . import hierclust
from . import flatclust
from .hierclust import hier_nmf2, TreeResults

__all__ = ["DenseMatrix", "SparseMatrix", "nmf_sparse", "load_matrix_market", "NmfSolver", "NmfResult", "nmf", "initialize", "finalize", "is_initialized",
           "make_options", "uniform_host", "nnls_blockpivot", "nmf_sharded", "Comm", "set_stream", "thread_context_begin", "thread_context_end", "trim_device_cache", "SmallkAPI", "hierclust", "flatclust", "hier_nmf2", "TreeResults", "_lib"]


def __getattr__(name):
    """``smallk_amd.Hierclust`` / ``smallk_amd.Flatclust`` / ``smallk_amd.pyclust`` (pysmallk's clustering classes,
    pysmallk/interface/smallk_lib.pyx:924-1420), loaded on first use."""
    if name in ("Hierclust", "Flatclust", "pyclust"):
        import importlib
        mod = importlib.import_module(".pyclust", __name__)
        return mod if name == "pyclust" else getattr(mod, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
