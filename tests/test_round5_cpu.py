"""CPU-side checks of round 5's additions: the oracle's planted-data generator against its defining formula and its own
sub-blocks, the synthetic sparse generators of the bench workloads, the bench's sparse entry points (no GPU call)."""
import os
import subprocess
import sys

import numpy as np

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planted_generator_is_its_formula():
    """orc_fill_planted = fp32(Ws Hs + noise U) with the thresholded uniform factors of seed + 1 / seed + 2."""
    m, n, ks, seed = 257, 130, 70, 11
    A = oracle.fill_planted(m, n, seed, ks)
    W = oracle.fill_uniform(m, ks, seed + 1)
    H = oracle.fill_uniform(ks, n, seed + 2)
    U = oracle.fill_uniform(m, n, seed)
    W = np.where(W > np.float32(0.7), W, 0.0)
    H = np.where(H > np.float32(0.7), H, 0.0)
    ref = (W @ H + 0.05 * U).astype(np.float32).astype(np.float64)
    assert np.max(np.abs(A - ref)) <= 2e-7 * np.max(ref)           # BLAS order vs the fixed fma chain: at most an fp32 ulp
    assert (A != ref).mean() < 1e-3
    # rank structure: the noise-free part has rank <= ks
    B = oracle.fill_planted(m, n, seed, 5, noise=0.0)
    assert np.linalg.matrix_rank(B, tol=1e-5) <= 5
    # bf16 storage rounds the same values
    assert np.array_equal(oracle.fill_planted(m, n, seed, ks, quant=1), oracle.quantize(A, 1))


def test_planted_generator_blocks_agree_with_the_whole():
    m, n, ks, seed = 300, 200, 33, 5
    A = oracle.fill_planted(m, n, seed, ks)
    for (r0, rows, c0, cols) in ((0, 300, 17, 1), (123, 1, 0, 200), (40, 50, 60, 70)):
        blk = oracle.fill_planted(rows, cols, seed, ks, r0=r0, c0=c0, gheight=m)
        assert np.array_equal(blk, A[r0:r0 + rows, c0:c0 + cols])


def test_term_document_generator():
    from smallk_amd.synthetic import term_document, community_graph
    A = term_document(3000, 2000, 60_000, seed=3)
    assert A.shape == (3000, 2000) and 0.8 * 60_000 < A.nnz < 1.3 * 60_000
    rows = np.diff(A.tocsr().indptr)
    cols = np.diff(A.indptr)
    assert rows.min() >= 1 and cols.min() >= 1                       # block pivoting needs non-singular Gram matrices
    assert rows.max() > 20 * np.median(rows)                         # Zipf: the skew the segment kernel exists for
    assert A.data.min() >= 1.0 and A.has_sorted_indices
    B = term_document(3000, 2000, 60_000, seed=3)
    assert (A != B).nnz == 0                                         # deterministic
    G, comm = community_graph(5000, 16, 16, seed=0)
    assert (G != G.T).nnz == 0 and G.shape == (5000, 5000) and comm.shape == (5000,)


def test_bench_knows_the_sparse_workloads_without_touching_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for w in ("s_reuters", "s_reuters_hals", "s_1m", "--data", "--single-copy"):
        assert w in r.stdout
    import bench_sparse
    A = bench_sparse.make_matrix("s_reuters")
    assert A.shape == (12411, 7984) and 4.0e5 < A.nnz < 6.0e5


def test_planted_generator_bits_are_pinned():
    """the committed block (tests/golden/planted_generator.npz, make_golden.make_planted_fixture): a change of the generator's
    formula or order of operations would silently change every planted-data measurement"""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "planted_generator.npz"))
    m, n, seed, ks = (int(v) for v in fx["params"])
    assert np.array_equal(oracle.fill_planted(m, n, seed, ks), fx["f32"])
    assert np.array_equal(oracle.fill_planted(m, n, seed, ks, quant=1), fx["bf16"])
