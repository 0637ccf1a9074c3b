"""CPU tests: the C-ABI library loads and exports every symbol include/smallk_amd.h declares;
host-side logic that needs no GPU (validation, generator, CSV, facade setters)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "smallk_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(smk_[a-z0-9_]+)\s*\(", text))
    names.discard("smk_allreduce_fn")
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import smallk_amd
    lib = C.CDLL(smallk_amd._lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) > 60
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/smallk_amd.h but not exported"
    # and the Python binding table covers the header
    assert set(names) <= set(smallk_amd._lib.SYMBOLS), set(names) - set(smallk_amd._lib.SYMBOLS)


def test_product_fails_loudly_without_gpu_or_library():
    import smallk_amd
    L = smallk_amd._lib
    if L.lib().smk_is_initialized() == L.INITIALIZED:
        pytest.skip("GPU present")
    o = smallk_amd.make_options(8, 8, 2, "MU")
    A = np.ones((8, 8), order="F")
    r = smallk_amd.nmf(A, np.ones((8, 2)), np.ones((2, 8)), "MU")
    assert r.result == L.NOTINITIALIZED          # Nmf() before NmfInitialize(): nmf.cpp:179-184
    h = C.c_void_p()
    assert L.lib().smk_matrix_create(C.byref(h), 8, 8, 0, 8, 0) == L.NOTINITIALIZED
    assert o.k == 2


def test_is_valid_matches_reference_rules():
    import smallk_amd
    L = smallk_amd._lib
    ok = smallk_amd.make_options(10, 8, 4, "BPP")
    assert L.lib().smk_is_valid(C.byref(ok), 1) == 1
    bad = [dict(k=0), dict(k=9), dict(tol=0.0), dict(tol=1.0), dict(min_iter=0), dict(max_iter=0), dict(tolcount=0)]
    for b in bad:
        kw = dict(min_iter=5, max_iter=10, tol=0.01, tolcount=1)
        k = b.pop("k", 4)
        kw.update(b)
        o = smallk_amd.make_options(10, 8, k, "BPP", **kw)
        assert L.lib().smk_is_valid(C.byref(o), 1) == 0, b
    # k > width is only a matrix check
    o = smallk_amd.make_options(10, 8, 9, "BPP")
    assert L.lib().smk_is_valid(C.byref(o), 0) == 1
    o = smallk_amd.make_options(10, 8, 3, "RANK2")
    assert L.lib().smk_is_valid(C.byref(o), 1) == 0      # RANK2 requires k == 2


def test_host_generator_equals_oracle():
    import oracle
    import smallk_amd
    for quant in (0, 1):
        a = smallk_amd.uniform_host(33, 9, 42, quant=quant)
        b = oracle.fill_uniform(33, 9, 42, quant=quant)
        assert np.array_equal(a, b)


def test_csv_roundtrip_and_reference_bytes(tmp_path):
    """w.csv/h.csv writer and init-file reader: byte-compare with the reference's own
    WriteDelimitedFile / LoadDelimitedFile (oracle/_ref, compiled from /root/reference)."""
    import smallk_amd
    L = smallk_amd._lib.lib()
    rng = np.random.default_rng(0)
    M = np.asfortranarray(rng.standard_normal((7, 5)) * 10.0 ** rng.integers(-8, 8, (7, 5)))
    mine = tmp_path / "mine.csv"
    for prec in (1, 4, 6, 17):
        assert L.smk_write_csv(M.ctypes.data_as(C.POINTER(C.c_double)), 7, 7, 5, str(mine).encode(), prec) == 1
        out = np.zeros((7, 5), order="F")
        h, w = C.c_uint(0), C.c_uint(0)
        assert L.smk_load_csv(str(mine).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), 35, C.byref(h), C.byref(w)) == 1
        assert (h.value, w.value) == (7, 5)
        assert np.allclose(out, M, rtol=10.0 ** (-prec + 1) if prec < 17 else 1e-15)
        ref_so = os.path.join(ROOT, "oracle", "_ref", "libref_csv.so")
        if os.path.exists(ref_so):
            ref = C.CDLL(ref_so)
            theirs = tmp_path / "theirs.csv"
            assert ref.ref_write_csv(M.ctypes.data_as(C.POINTER(C.c_double)), 7, 7, 5, str(theirs).encode(), prec) == 1
            assert mine.read_bytes() == theirs.read_bytes()
            out2 = np.zeros((7, 5), order="F")
            assert ref.ref_load_csv(str(mine).encode(), out2.ctypes.data_as(C.POINTER(C.c_double)), C.c_ulong(35),
                                    C.byref(h), C.byref(w)) == 1
            assert np.array_equal(out, out2)
    # leading comment / blank lines are skipped (delimited_file.cpp:35-70)
    f = tmp_path / "c.csv"
    f.write_text("# comment\n\n% another\n1,2\n3,4\n")
    out = np.zeros((2, 2), order="F")
    h, w = C.c_uint(0), C.c_uint(0)
    assert L.smk_load_csv(str(f).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), 4, C.byref(h), C.byref(w)) == 1
    assert np.array_equal(out, np.array([[1.0, 2.0], [3.0, 4.0]]))


def test_facade_setters_getters_and_clamping(tmp_path):
    """smallk:: parameter semantics (smallk/src/smallk.cpp:391-468; smallk_test.cpp:56-110)."""
    import smallk_amd
    l = smallk_amd._lib.lib()
    l.smk_api_reset()
    assert l.smk_api_get_min_iter() == 5 and l.smk_api_get_max_iter() == 5000
    assert l.smk_api_get_nmf_tolerance() == 0.005 and l.smk_api_get_output_precision() == 6
    assert l.smk_api_get_output_format() == 1 and l.smk_api_get_max_terms() == 5
    l.smk_api_set_max_iter(0); assert l.smk_api_get_max_iter() == 1
    l.smk_api_set_min_iter(0); assert l.smk_api_get_min_iter() == 1
    l.smk_api_set_output_precision(0); assert l.smk_api_get_output_precision() == 1
    l.smk_api_set_output_precision(99); assert l.smk_api_get_output_precision() == 17
    l.smk_api_set_max_threads(0); assert l.smk_api_get_max_threads() == 1
    l.smk_api_set_max_threads(10 ** 6); assert l.smk_api_get_max_threads() == (os.cpu_count() or 2)
    assert l.smk_api_set_nmf_tolerance(0.0) == 1 and l.smk_api_set_nmf_tolerance(1.0) == 1     # logic_error
    assert l.smk_api_set_nmf_tolerance(0.25) == 0 and l.smk_api_get_nmf_tolerance() == 0.25
    assert l.smk_api_set_output_dir(str(tmp_path / "nope").encode()) == 1                       # logic_error
    assert l.smk_api_set_output_dir(str(tmp_path).encode()) == 0
    assert l.smk_api_get_output_dir().decode() == str(tmp_path) + "/"
    # Nmf() guards (smallk.cpp:476-492)
    assert l.smk_api_nmf(4, 1, b"", b"") == 1 and b"no matrix" in l.smk_api_last_exception()
    A = np.asfortranarray(np.arange(12.0).reshape(3, 4))
    assert l.smk_api_load_matrix_dense(A.ctypes.data_as(C.POINTER(C.c_double)), 3, 3, 4) == 0
    assert l.smk_api_is_matrix_loaded() == 1
    assert l.smk_api_nmf(0, 1, b"", b"") == 1 and b"k must be greater" in l.smk_api_last_exception()
    l.smk_api_set_min_iter(10); l.smk_api_set_max_iter(5)
    assert l.smk_api_nmf(2, 1, b"", b"") == 1 and b"min_iterations exceeds" in l.smk_api_last_exception()
    l.smk_api_reset()
    assert l.smk_api_is_matrix_loaded() == 0
    assert l.smk_api_get_major_version() == 1 and l.smk_api_get_minor_version() == 6 and l.smk_api_get_patch_level() == 2
