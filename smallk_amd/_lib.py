"""ctypes binding of libsmallk_amd.so (the C ABI in include/smallk_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C smallk_amd/csrc``.
There is NO CPU fallback: if the shared object is missing, importing the product
fails loudly; if no GPU is present, ``initialize()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMK_LIB_PATH") or os.path.join(_HERE, "lib", "libsmallk_amd.so")      # SMK_LIB_PATH: an A/B build of the library (tools/)

# Result codes (include/smallk_amd.h)
OK, NOTINITIALIZED, INITIALIZED, BAD_PARAM, FAILURE, SIZE_TOO_LARGE = 0, -1, -2, -3, -4, -5
DEVICE_ERROR, UNSUPPORTED = -100, -101
ALG_MU, ALG_HALS, ALG_RANK2, ALG_BPP = 0, 1, 2, 3
PROG_PG_RATIO, PROG_DELTA_FNORM = 0, 1
STORE_F32, STORE_BF16 = 0, 1

RESULT_NAMES = {0: "OK", -1: "NOTINITIALIZED", -2: "INITIALIZED", -3: "BAD_PARAM", -4: "FAILURE",
                -5: "SIZE_TOO_LARGE", -6: "FLATCLUST_FAILURE", -100: "DEVICE_ERROR", -101: "UNSUPPORTED"}


class Options(C.Structure):
    """struct smk_options == NmfOptions (common/include/nmf.hpp:55-69)."""
    _fields_ = [("tol", C.c_double), ("algorithm", C.c_int), ("prog_est_algorithm", C.c_int),
                ("height", C.c_int), ("width", C.c_int), ("k", C.c_int),
                ("min_iter", C.c_int), ("max_iter", C.c_int), ("tolcount", C.c_int),
                ("max_threads", C.c_int), ("verbose", C.c_int), ("normalize", C.c_int)]


class Stats(C.Structure):
    """struct smk_stats == NmfStats (common/include/nmf.hpp:43-53)."""
    _fields_ = [("elapsed_us", C.c_ulonglong), ("iteration_count", C.c_int)]


class ClustOptions(C.Structure):
    """struct smk_clust_options == ClustOptions (hierclust/include/clust.hpp:27-37)."""
    _fields_ = [("nmf", Options), ("maxterms", C.c_int), ("unbalanced", C.c_double), ("trial_allowance", C.c_int),
                ("num_clusters", C.c_int), ("verbose", C.c_int), ("flat", C.c_int)]


class ClustStats(C.Structure):
    """struct smk_clust_stats == ClustStats (hierclust/include/clust.hpp:20-25)."""
    _fields_ = [("nmf_count", C.c_int), ("max_count", C.c_int)]


class TreeNodeInfo(C.Structure):
    """struct smk_tree_node: the scalar fields of TreeNode<T> (hierclust/include/tree.hpp:31-50)."""
    _fields_ = [("priority", C.c_double), ("parent", C.c_uint), ("left_child", C.c_uint), ("right_child", C.c_uint),
                ("is_valid", C.c_int), ("is_left_child", C.c_int), ("is_leaf", C.c_int), ("doc_count", C.c_int64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

# every symbol include/smallk_amd.h declares: name -> (restype, argtypes)
_dp = C.POINTER(C.c_double)
_i64 = C.c_int64
_vp = C.c_void_p
SYMBOLS = {
    "smk_initialize": (C.c_int, [C.c_int]),
    "smk_is_initialized": (C.c_int, []),
    "smk_finalize": (None, []),
    "smk_last_error": (C.c_char_p, []),
    "smk_device_cu_count": (C.c_int, []),
    "smk_is_valid": (C.c_int, [C.POINTER(Options), C.c_int]),
    "smk_set_stream": (C.c_int, [_vp]),
    "smk_nmf_dense": (C.c_int, [C.POINTER(Options), _dp, _i64, _dp, _i64, _dp, _i64, C.POINTER(Stats), C.c_int]),
    "smk_matrix_create": (C.c_int, [C.POINTER(_vp), _i64, _i64, _i64, _i64, C.c_int]),
    "smk_matrix_upload_f64": (C.c_int, [_vp, _dp, _i64]),
    "smk_matrix_create_single_copy": (C.c_int, [C.POINTER(_vp), _i64, _i64, _i64, _i64, C.c_int]),
    "smk_matrix_is_single_copy": (C.c_int, [_vp]),
    "smk_matrix_device_bytes": (_i64, [_vp]),
    "smk_matrix_fill_uniform": (C.c_int, [_vp, C.c_uint64]),
    "smk_matrix_fill_planted": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_double, C.c_double]),
    "smk_matrix_download_f64": (C.c_int, [_vp, _dp, _i64]),
    "smk_matrix_destroy": (None, [_vp]),
    "smk_matrix_clone": (C.c_int, [_vp, C.POINTER(_vp)]),
    "smk_thread_context_begin": (C.c_int, [C.c_int]),
    "smk_thread_context_end": (None, []),
    "smk_device_count": (C.c_int, []),
    "smk_current_device": (C.c_int, []),
    "smk_device_synchronize": (C.c_int, []),
    "smk_matrix_create_sparse": (C.c_int, [C.POINTER(_vp), _i64, _i64, _i64, _i64, _i64, C.POINTER(C.c_uint),
                                           C.POINTER(C.c_uint), _dp]),
    "smk_nmf_sparse": (C.c_int, [C.POINTER(Options), C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_uint),
                                 C.POINTER(C.c_uint), _dp, _dp, _i64, _dp, _i64, C.POINTER(Stats)]),
    "smk_load_matrix_market": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                         C.POINTER(C.c_uint), C.POINTER(C.c_uint), _dp]),
    "smk_csc_transpose": (C.c_int, [_i64, _i64, C.POINTER(C.c_uint), C.POINTER(C.c_uint), _dp, C.POINTER(C.c_uint),
                                    C.POINTER(C.c_uint), _dp]),
    "smk_csc_subset_cols_compact": (C.c_int, [_i64, _i64, C.POINTER(C.c_uint), C.POINTER(C.c_uint), _dp,
                                              C.POINTER(C.c_uint), _i64, C.POINTER(C.c_uint), C.POINTER(C.c_uint), _dp,
                                              C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(_i64), C.POINTER(_i64)]),
    "smk_matrix_download_csc": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint), _dp]),
    "smk_device_trim": (C.c_size_t, []),
    "smk_matrix_nnz": (_i64, [_vp]),
    "smk_matrix_sparse_product": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _i64, _dp, _i64, C.c_int, C.POINTER(C.c_double)]),
    "smk_matrix_height": (_i64, [_vp]),
    "smk_uniform_fill_host": (None, [_dp, _i64, _i64, _i64, _i64, _i64, _i64, C.c_uint64, C.c_int]),
    "smk_solver_create": (C.c_int, [C.POINTER(_vp), C.POINTER(Options), _vp]),
    "smk_solver_destroy": (None, [_vp]),
    "smk_solver_set_factors": (C.c_int, [_vp, _dp, _i64, _dp, _i64]),
    "smk_solver_set_factors_uniform": (C.c_int, [_vp, C.c_uint64, C.c_uint64]),
    "smk_solver_run": (C.c_int, [_vp, C.POINTER(Stats)]),
    "smk_solver_iterate": (C.c_int, [_vp, C.c_int]),
    "smk_solver_iterate_checked": (C.c_int, [_vp, C.c_int, _dp]),
    "smk_solver_sync": (C.c_int, [_vp]),
    "smk_solver_progress": (C.c_int, [_vp, _dp]),
    "smk_solver_get_factors": (C.c_int, [_vp, C.c_int, _dp, _i64, _dp, _i64]),
    "smk_solver_iteration_count": (C.c_int, [_vp]),
    "smk_solver_product_form": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp]),
    "smk_nnls_blockpivot": (C.c_int, [C.c_int, _i64, _dp, _i64, _dp, _i64, _dp, _i64, _dp, _i64]),
    "smk_solver_enable_timing": (C.c_int, [_vp, C.c_int]),
    "smk_solver_kernel_time": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(C.c_int)]),
    "smk_solver_kernel_work": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "smk_debug_nnls_stats": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "smk_solver_kernel_name": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int]),
    "smk_solver_comm_workspace_bytes": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "smk_comm_unique_id": (C.c_int, [_vp]),
    "smk_comm_init_rank": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int]),
    "smk_comm_init_all": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(C.c_int)]),
    "smk_comm_init_local": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "smk_comm_selftest": (C.c_int, [_vp]),
    "smk_comm_rank": (C.c_int, [_vp]),
    "smk_comm_world": (C.c_int, [_vp]),
    "smk_comm_destroy": (None, [_vp]),
    "smk_comm_abort": (None, [_vp]),
    "smk_solver_attach_comm": (C.c_int, [_vp, _vp]),
    "smk_nmf_dense_sharded": (C.c_int, [C.POINTER(Options), _dp, _i64, _dp, _i64, _dp, _i64, C.POINTER(Stats), C.c_int,
                                        C.c_int, C.POINTER(C.c_int), C.c_int]),
    "smk_solver_set_comm": (C.c_int, [_vp, C.c_int, C.c_int, ALLREDUCE_FN, _vp, _vp, C.c_size_t]),
    # CSV helpers (facade.cpp; reference delimited_file.hpp:49-135)
    "smk_write_csv": (C.c_int, [_dp, C.c_uint, C.c_uint, C.c_uint, C.c_char_p, C.c_uint]),
    "smk_load_csv": (C.c_int, [C.c_char_p, _dp, C.c_ulong, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    # flat handles onto namespace smallk (facade.cpp; pysmallk/interface/smallk_lib.pyx:42-88)
    "smk_api_last_exception": (C.c_char_p, []),
    "smk_api_initialize": (C.c_int, []),
    "smk_api_is_initialized": (C.c_int, []),
    "smk_api_finalize": (None, []),
    "smk_api_reset": (None, []),
    "smk_api_seed_rng": (None, [C.c_int]),
    "smk_api_get_major_version": (C.c_uint, []),
    "smk_api_get_minor_version": (C.c_uint, []),
    "smk_api_get_patch_level": (C.c_uint, []),
    "smk_api_load_matrix_file": (C.c_int, [C.c_char_p]),
    "smk_api_load_matrix_dense": (C.c_int, [_dp, C.c_uint, C.c_uint, C.c_uint]),
    "smk_api_load_matrix_sparse": (C.c_int, [C.c_uint, C.c_uint, C.c_uint, _dp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "smk_api_is_matrix_loaded": (C.c_int, []),
    "smk_api_set_output_dir": (C.c_int, [C.c_char_p]),
    "smk_api_get_output_dir": (C.c_char_p, []),
    "smk_api_set_output_precision": (None, [C.c_uint]),
    "smk_api_get_output_precision": (C.c_uint, []),
    "smk_api_set_nmf_tolerance": (C.c_int, [C.c_double]),
    "smk_api_get_nmf_tolerance": (C.c_double, []),
    "smk_api_set_max_iter": (None, [C.c_uint]),
    "smk_api_get_max_iter": (C.c_uint, []),
    "smk_api_set_min_iter": (None, [C.c_uint]),
    "smk_api_get_min_iter": (C.c_uint, []),
    "smk_api_set_max_threads": (None, [C.c_uint]),
    "smk_api_get_max_threads": (C.c_uint, []),
    "smk_api_set_max_terms": (None, [C.c_uint]),
    "smk_api_get_max_terms": (C.c_uint, []),
    "smk_api_set_output_format": (None, [C.c_int]),
    "smk_api_get_output_format": (C.c_int, []),
    "smk_api_set_hiernmf2_tolerance": (C.c_int, [C.c_double]),
    "smk_api_get_hiernmf2_tolerance": (C.c_double, []),
    "smk_api_set_device_storage": (None, [C.c_int]),
    "smk_api_get_device_storage": (C.c_int, []),
    "smk_api_get_iteration_count": (C.c_uint, []),
    "smk_api_nmf": (C.c_int, [C.c_uint, C.c_int, C.c_char_p, C.c_char_p]),
    "smk_api_locked_buffer_w": (_dp, [C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "smk_api_locked_buffer_h": (_dp, [C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "smk_api_hiernmf2": (C.c_int, [C.c_uint]),
    "smk_api_hiernmf2_with_flat": (C.c_int, [C.c_uint]),
    "smk_api_load_dictionary_file": (C.c_int, [C.c_char_p]),
    "smk_api_load_dictionary": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint]),
    # HierNMF2 (hierclust.cpp; reference hierclust/include/clust.hpp, tree.hpp)
    "smk_clust_is_valid": (C.c_int, [C.POINTER(ClustOptions), C.c_int]),
    "smk_matrix_gather_cols": (C.c_int, [_vp, C.POINTER(C.c_uint), _i64, C.POINTER(_vp), C.POINTER(C.c_uint),
                                         C.POINTER(_i64)]),
    "smk_clust_dense": (C.c_int, [C.POINTER(ClustOptions), _dp, _i64, C.c_int, C.c_uint64, C.POINTER(C.c_uint64),
                                  C.c_char_p, C.POINTER(_vp), C.POINTER(ClustStats)]),
    "smk_clust_sparse": (C.c_int, [C.POINTER(ClustOptions), _i64, C.POINTER(C.c_uint), C.POINTER(C.c_uint), _dp,
                                   C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.POINTER(_vp),
                                   C.POINTER(ClustStats)]),
    "smk_clust_resident": (C.c_int, [C.POINTER(ClustOptions), _vp, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p,
                                     C.POINTER(_vp), C.POINTER(ClustStats)]),
    "smk_tree_destroy": (None, [_vp]),
    "smk_tree_node_count": (C.c_int, [_vp]),
    "smk_tree_term_count": (_i64, [_vp]),
    "smk_tree_doc_count": (_i64, [_vp]),
    "smk_tree_get_node": (C.c_int, [_vp, C.c_int, C.POINTER(TreeNodeInfo)]),
    "smk_tree_node_docs": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint)]),
    "smk_tree_node_topic": (C.c_int, [_vp, C.c_int, _dp]),
    "smk_tree_node_terms": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "smk_tree_assignments": (_i64, [_vp, C.POINTER(C.c_uint)]),
    "smk_tree_outliers": (_i64, [_vp, C.POINTER(C.c_uint)]),
    "smk_tree_write_assignments": (C.c_int, [_vp, C.c_char_p]),
    "smk_tree_write": (C.c_int, [_vp, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), _i64]),
    "smk_clust_priority": (C.c_double, [_dp, _dp, _i64]),
    "smk_tree_flat_factors": (C.c_int, [_vp, _dp, _i64, _dp, _i64]),
    # flat clustering (flatclust.cpp; reference flatclust/src/flat_clust.cpp, common/include/assignments.hpp)
    "smk_flatclust_dense": (C.c_int, [C.POINTER(Options), _dp, _i64, _dp, _i64, _dp, _i64, C.POINTER(Stats), C.c_int]),
    "smk_flatclust_sparse": (C.c_int, [C.POINTER(Options), C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_uint),
                                       C.POINTER(C.c_uint), _dp, _dp, _i64, _dp, _i64, C.POINTER(Stats)]),
    "smk_solver_nnls_hals": (C.c_int, [_vp, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "smk_compute_assignments": (C.c_int, [_dp, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_uint)]),
    "smk_compute_fuzzy_assignments": (C.c_int, [_dp, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_float)]),
    "smk_top_terms": (C.c_int, [C.c_int, _dp, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_int)]),
    "smk_write_assignments_file": (C.c_int, [C.POINTER(C.c_uint), C.c_uint, C.c_char_p]),
    "smk_write_fuzzy_assignments_file": (C.c_int, [C.POINTER(C.c_float), C.c_uint, C.c_uint, C.c_char_p]),
    "smk_flatclust_write_results": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint), C.c_uint,
                                              C.POINTER(C.c_float), C.POINTER(C.c_char_p), _i64, C.POINTER(C.c_int),
                                              _i64, C.c_int, C.c_uint, C.c_uint, C.c_uint]),
}

_lib = None


def _preload_torch_hip_runtime():
    """A process must not end up with two HIP runtimes.  PyTorch-ROCm wheels bundle their own
    libamdhip64/libhsa-runtime64; if torch is installed, load those first (RTLD_GLOBAL) so that
    libsmallk_amd.so binds to the same runtime no matter which of the two is imported first.
    Without torch the system runtime under /opt/rocm is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if not spec or not spec.origin:
        return
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(d, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


class SmallkError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {RESULT_NAMES.get(code, code)}" + (f" ({detail})" if detail else ""))


def lib():
    """Load the shared library (no compute, no GPU needed for loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C smallk_amd/csrc`.  There is no CPU fallback.")
        _preload_torch_hip_runtime()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, where):
    if rc != OK:
        detail = lib().smk_last_error()
        raise SmallkError(rc, where, detail.decode() if detail else "")
    return rc
