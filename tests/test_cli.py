"""The `nmf` command line tool (reference: nmf/src/main.cpp, regression shape of tests/scripts/test_nmf.sh:
fixed init files + --miniter 1, then compare w.csv / h.csv)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NMF = os.path.join(ROOT, "smallk_amd", "bin", "nmf")


def write_csv(path, M, prec=17):
    import smallk_amd
    M = np.asfortranarray(M, dtype=np.float64)
    assert smallk_amd._lib.lib().smk_write_csv(M.ctypes.data_as(C.POINTER(C.c_double)), M.shape[0], M.shape[0],
                                               M.shape[1], str(path).encode(), prec) == 1


def test_cli_usage_and_argument_errors():
    assert os.path.exists(NMF), "build the CLI with make -C smallk_amd/csrc"
    r = subprocess.run([NMF], capture_output=True, text=True)
    assert r.returncode == 0 and "--matrixfile" in r.stdout                       # no args -> help
    r = subprocess.run([NMF, "--k", "4"], capture_output=True, text=True)
    assert r.returncode != 0 and "required command line argument --matrixfile" in r.stderr
    r = subprocess.run([NMF, "--matrixfile", "x.csv"], capture_output=True, text=True)
    assert r.returncode != 0 and "required command line argument --k" in r.stderr
    r = subprocess.run([NMF, "--matrixfile", "x.csv", "--k", "4", "--algorithm", "FOO"], capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid value" in r.stderr
    r = subprocess.run([NMF, "--matrixfile", "x.csv", "--k", "4", "--tol", "2.0"], capture_output=True, text=True)
    assert r.returncode != 0 and "tolerance must be in the interval" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("alg,stopping", [("BPP", "PG_RATIO"), ("HALS", "PG_RATIO"), ("MU", "DELTA"), ("RANK2", "PG_RATIO")])
def test_cli_dense_csv(tmp_path, alg, stopping):
    m, n, k = 256, 192, (2 if alg == "RANK2" else 8)
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    write_csv(tmp_path / "a.csv", A)
    write_csv(tmp_path / "w0.csv", W0)
    write_csv(tmp_path / "h0.csv", H0)
    r = subprocess.run([NMF, "--matrixfile", str(tmp_path / "a.csv"), "--k", str(k), "--algorithm", alg,
                        "--stopping", stopping, "--infile_W", str(tmp_path / "w0.csv"), "--infile_H", str(tmp_path / "h0.csv"),
                        "--outfile_W", str(tmp_path / "w.csv"), "--outfile_H", str(tmp_path / "h.csv"),
                        "--miniter", "1", "--maxiter", "500", "--tol", "0.01", "--outprecision", "10", "--verbose", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ref = oracle.nmf(A, W0, H0, alg, min_iter=1, max_iter=500, tol=0.01,
                     prog_est=oracle.DELTA_FNORM if stopping == "DELTA" else oracle.PG_RATIO)
    W = np.loadtxt(tmp_path / "w.csv", delimiter=",", ndmin=2)
    H = np.loadtxt(tmp_path / "h.csv", delimiter=",", ndmin=2)
    assert W.shape == (m, k) and H.shape == (k, n)
    assert np.linalg.norm(W - ref.W) / np.linalg.norm(ref.W) < 1e-4
    assert np.linalg.norm(H - ref.H) / np.linalg.norm(ref.H) < 1e-4
    first = open(tmp_path / "w.csv").readline().strip().split(",")
    assert len(first) == k and all("e" in t and len(t.split(".")[1].split("e")[0]) == 10 for t in first)   # scientific, 10 digits


@pytest.mark.gpu
def test_cli_sparse_matrix_market(tmp_path):
    m, n, k = 300, 220, 6
    rng = np.random.default_rng(5)
    A = sp.random(m, n, density=0.2, random_state=rng, data_rvs=lambda s: rng.random(s) + 0.05, format="coo")
    with open(tmp_path / "a.mtx", "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%% generated\n")
        f.write(f"{m} {n} {A.nnz}\n")
        for r_, c_, v in zip(A.row, A.col, A.data):
            f.write(f"{r_ + 1} {c_ + 1} {float(v)!r}\n")
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    write_csv(tmp_path / "w0.csv", W0)
    write_csv(tmp_path / "h0.csv", H0)
    r = subprocess.run([NMF, "--matrixfile", str(tmp_path / "a.mtx"), "--k", str(k), "--infile_W", str(tmp_path / "w0.csv"),
                        "--infile_H", str(tmp_path / "h0.csv"), "--outfile_W", str(tmp_path / "w.csv"),
                        "--outfile_H", str(tmp_path / "h.csv"), "--miniter", "1", "--tol", "0.01", "--outprecision", "12"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "progress metric" in r.stdout and "converged after" in r.stdout          # verbose defaults to 1
    ref = oracle.nmf(A.toarray(), W0, H0, "BPP", min_iter=1, max_iter=5000, tol=0.01)
    W = np.loadtxt(tmp_path / "w.csv", delimiter=",")
    H = np.loadtxt(tmp_path / "h.csv", delimiter=",")
    assert np.linalg.norm(W - ref.W) < 1e-7 and np.linalg.norm(H - ref.H) < 1e-7
