"""The reference's OWN caller programs running on this library, unmodified.

`make -C oracle ref` compiles smallk/src/smallk_test.cpp and examples/smallk_example.cpp where they lie under
/root/reference against this repo's include/smallk.hpp and links them with libsmallk_amd.so (oracle/_ref/ is
git-ignored but travels to the GPU box).  The programs read reuters.mtx / reuters_dictionary.txt /
init files from a data directory the reference keeps in a separate repository; here that directory
is synthesised (planted-topic term-document matrix of the same shape class)."""
import os
import subprocess

import numpy as np
import pytest

from hier_cases import planted
from test_cli import write_csv
from test_cli_clust import _write_mtx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_BIN = os.path.join(ROOT, "oracle", "_ref", "smallk_test")
EXAMPLE_BIN = os.path.join(ROOT, "oracle", "_ref", "smallk_example")


def test_reference_sources_compile_against_our_headers(tmp_path):
    """Source-level drop-in: the reference programs need nothing but include/smallk.hpp + the library."""
    src = "/root/reference/smallk/src/smallk_test.cpp"
    if not os.path.exists(src):
        pytest.skip("reference tree absent")
    for s in (src, "/root/reference/examples/smallk_example.cpp"):
        out = tmp_path / (os.path.basename(s) + ".bin")
        r = subprocess.run(["g++", "-std=c++11", "-I" + os.path.join(ROOT, "include"), s, "-o", str(out),
                            "-L" + os.path.join(ROOT, "smallk_amd", "lib"), "-lsmallk_amd"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


def _data_dir(tmp_path, m, n, topics, k_init, wname, hname):
    import oracle
    d = tmp_path / "data"
    d.mkdir()
    A, _ = planted(m, n, topics, 77, sparse=True)
    _write_mtx(d / "reuters.mtx", A)
    dictionary = [f"term{i}" for i in range(m)]
    (d / "reuters_dictionary.txt").write_text("\n".join(dictionary) + "\n")
    W0, H0 = oracle.fill_uniform(m, k_init, 5), oracle.fill_uniform(k_init, n, 6)
    write_csv(d / wname, W0)
    write_csv(d / hname, H0)
    return d, A, W0, H0, dictionary


@pytest.mark.gpu
def test_reference_smallk_test_program(tmp_path):
    """smallk/src/smallk_test.cpp:56-146: setters/getters, Reset, LoadMatrix(.mtx), Nmf(8, BPP, init files) with
    MinIter 1, LoadDictionary, HierNmf2(5)."""
    import oracle
    if not os.path.exists(TEST_BIN):
        pytest.skip("oracle/_ref/smallk_test not built (reference tree absent at build time)")
    m, n = 300, 420
    d, A, W0, H0, dictionary = _data_dir(tmp_path, m, n, 6, 8, "nmf_init_w.csv", "nmf_init_h.csv")
    run = tmp_path / "run"
    run.mkdir()
    r = subprocess.run([TEST_BIN, str(d)], cwd=run, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SMALLK_SEED="12345"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.strip() == "" or "amdgpu.ids" in r.stderr, r.stderr[-2000:]      # the program prints exceptions to stderr
    assert "Running NMF-BPP" in r.stdout and "Running HierNmf2" in r.stdout
    # Nmf(8, BPP, w, h): defaults after Reset() are tol 0.005, max_iter 5000; MinIter 1, precision 6
    ref = oracle.nmf(A.toarray(), W0, H0, "BPP", min_iter=1, max_iter=5000, tol=0.005)
    W = np.loadtxt(run / "w.csv", delimiter=",", ndmin=2)
    H = np.loadtxt(run / "h.csv", delimiter=",", ndmin=2)
    assert W.shape == (m, 8) and H.shape == (8, n)
    assert np.linalg.norm(W - ref.W) / np.linalg.norm(ref.W) < 1e-5           # 6 printed digits
    assert np.linalg.norm(H - ref.H) / np.linalg.norm(ref.H) < 1e-5
    # HierNmf2(5): XML tree + assignments (random initialisers: check structure, not values)
    tree = (run / "tree_5.xml").read_text()
    assert tree.startswith('<?xml version="1.0"?>') and tree.count("<node id=") == 8
    labels = (run / "assignments_5.csv").read_text().splitlines()[0].split(",")
    assert len(labels) == n and {int(x) for x in labels} <= set(range(-1, 8))


@pytest.mark.gpu
def test_reference_example_program(tmp_path):
    """examples/smallk_example.cpp: Nmf(32) BPP, Nmf(16, HALS), Nmf(2, RANK2, init files) twice, LockedBufferW/H,
    HierNmf2(5) JSON, HierNmf2(10) XML with 12 terms, HierNmf2WithFlat(18)."""
    if not os.path.exists(EXAMPLE_BIN):
        pytest.skip("oracle/_ref/smallk_example not built (reference tree absent at build time)")
    # 40 planted topics: Nmf(32) needs a matrix of rank >= 32 or BPP meets a singular HH' (in the
    # reference as well)
    d, A, W0, H0, dictionary = _data_dir(tmp_path, 400, 600, 40, 2, "nmf_rank2_init_w.csv", "nmf_rank2_init_h.csv")
    run = tmp_path / "run"
    run.mkdir()
    # Initialize() seeds the RNG from the clock like the reference; Nmf(32) BPP from a random start fails on ~4 % of the
    # seeds on this rank-40 input (singular passive block), so the run is pinned to one seed
    r = subprocess.run([EXAMPLE_BIN, str(d)], cwd=run, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SMALLK_SEED="12345"))
    assert r.returncode == 0, r.stderr[-2000:]
    for name in ("w.csv", "h.csv", "assignments_5.csv", "tree_5.json", "assignments_10.csv", "tree_10.xml"):
        assert (run / name).exists(), (name, r.stdout[-1500:], r.stderr[-1500:])
    # the flat step needs 18 leaves; on this input the search may stop earlier, which the library
    # reports exactly like the reference (runtime_error printed by the program, files of the tree kept)
    if (run / "clusters_18.xml").exists():
        assert (run / "assignments_flat_18.csv").exists() and (run / "assignments_fuzzy_18.csv").exists()
    else:
        assert "Insufficient number of leaf nodes" in r.stderr
