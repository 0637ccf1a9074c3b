"""The gather product of sparse NMF at ranks 3 .. 128 (spmm_seg.hip) by itself, through the C ABI
(smk_matrix_sparse_product), against scipy on the same CSC -- the reference's sparse Gemm
(common/include/sparse_gemm_ab_impl.hpp:24-100,480-582; sparse_gemm_ba_impl.hpp:25-99).
Shapes the entry-balanced segments have to get right: empty columns and rows (leading, trailing, runs), columns longer
than a segment (pieces + fix-up), columns exactly a segment long, a single column, skewed term-document data."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _check(gpu, A, k, seed=0):
    rng = np.random.default_rng(seed)
    A = A.tocsc()
    A.sort_indices()
    m, n = A.shape
    S = gpu.SparseMatrix(A.data, A.indices, A.indptr, A.shape)
    X = np.asfortranarray(rng.random((k, m)))
    Y = np.asfortranarray(rng.random((k, n)))
    got = S.product(X)
    ref = (A.T @ X.T).T
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-13), (k, np.abs(got - ref).max())
    got = S.product(Y, transposed=True)
    ref = (A @ Y.T).T
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-13), (k, np.abs(got - ref).max())
    S.close()


@pytest.mark.parametrize("k", [3, 8, 16, 17, 32, 48, 64, 100, 128])
def test_random_with_empty_rows_and_columns(gpu, k):
    rng = np.random.default_rng(k)
    A = sp.random(700, 500, density=0.03, random_state=rng, format="lil")
    A[:, :7] = 0          # leading empties
    A[:, 200:260] = 0     # a run of empties longer than nothing
    A[:, -3:] = 0         # trailing empties
    A[:5, :] = 0
    A[300:340, :] = 0
    _check(gpu, A.tocsc(), k)


@pytest.mark.parametrize("k", [8, 32, 64, 128])
def test_long_columns_and_exact_segment_lengths(gpu, k):
    rng = np.random.default_rng(100 + k)
    m, n = 3000, 400
    A = sp.random(m, n, density=0.004, random_state=rng, format="lil")
    for j, ln in ((0, 64), (1, 65), (2, 63), (3, 128), (4, 129), (5, 1000), (6, 3000), (399, 777), (200, 64), (201, 64), (202, 1)):
        A[:, j] = 0
        rows = rng.choice(m, size=ln, replace=False)
        A[rows, j] = rng.random(ln) + 0.1
    _check(gpu, A.tocsc(), k)


def test_single_column_and_single_row(gpu):
    rng = np.random.default_rng(5)
    col = sp.csc_matrix(rng.random((500, 1)))
    _check(gpu, col, 16)
    row = sp.csc_matrix(rng.random((1, 500)))
    _check(gpu, row, 16)
    one = sp.csc_matrix(np.array([[0.0, 0.0], [0.0, 2.5], [0.0, 0.0]]))
    _check(gpu, one, 8)


@pytest.mark.parametrize("k", [16, 32])
def test_term_document_shape(gpu, k):
    from smallk_amd.synthetic import term_document
    A = term_document(12411, 7984, 500_000, seed=1)
    assert np.diff(A.indptr).min() >= 1 and np.diff(A.tocsr().indptr).min() >= 1
    assert np.diff(A.tocsr().indptr).max() > 2000        # the skew the segments exist for
    _check(gpu, A, k)


def test_segment_kernel_equals_round4_kernel_in_a_run(gpu, monkeypatch):
    """MU on skewed data with empty columns (MU tolerates them): the solver on segments against the oracle's dense run."""
    import oracle
    rng = np.random.default_rng(9)
    m, n, k = 900, 600, 24
    A = sp.random(m, n, density=0.02, random_state=rng, format="lil")
    A[:, 10:20] = 0
    A[50:60, :] = 0
    A[rng.choice(m, 300, replace=False), 5] = 1.0
    A[7, rng.choice(n, 400, replace=False)] = 2.0
    A = A.tocsc()
    W0, H0 = oracle.fill_uniform(m, k, 43), oracle.fill_uniform(k, n, 44)
    ref = oracle.nmf(A.toarray(), W0, H0, "MU", min_iter=8, max_iter=8)
    got = gpu.nmf_sparse(A, W0, H0, "MU", min_iter=8, max_iter=8)
    assert ref.result == 0 and got.result == 0
    assert np.linalg.norm(got.W - ref.W) < 1e-8 and np.linalg.norm(got.H - ref.H) < 1e-8


@pytest.mark.parametrize("alg,k", [("MU", 8), ("BPP", 24), ("HALS", 16)])
def test_column_sharded_sparse_run_on_the_segment_kernel(gpu, alg, k):
    """Sparse A column-sharded over two ranks (in-process stand-in communicator, both on this GPU): every rank's shard gets its
    own segment plans (CSC of the local columns and of their transpose); factors against the dense oracle on the whole matrix."""
    import threading
    import oracle
    from smallk_amd import Comm, NmfSolver, SparseMatrix, make_options, thread_context_begin, thread_context_end
    from smallk_amd import dist as sdist
    from smallk_amd.synthetic import term_document
    m, n, iters, world = 1200, 900, 6, 2
    A = term_document(m, n, 30_000, seed=4).tocsc()
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 * A.mean() / (0.5 * k) + 0.05)
    ref = oracle.nmf(A.toarray(), W0, H0, alg, min_iter=iters, max_iter=iters, normalize=False)
    assert ref.result == 0
    comms = Comm.init_local(world)
    out, errors = [None] * world, []

    def run(rank):
        try:
            thread_context_begin(0)
            c0, nc = sdist.shard_columns(n, world, rank)
            sub = A[:, c0:c0 + nc].tocsc()
            S = SparseMatrix(sub.data, sub.indices, sub.indptr, (m, nc), col0=c0, width_global=n)
            sv = NmfSolver(S, make_options(m, n, k, alg, min_iter=iters, max_iter=iters, normalize=False))
            sv.attach_comm(comms[rank])
            sv.set_factors(W0, H0[:, c0:c0 + nc])
            sv.iterate(iters)
            rc = sv.sync()
            W, H = sv.factors(normalize=False)
            out[rank] = (rc, W, H)
            sv.close()
            S.close()
        except Exception as e:          # pragma: no cover
            errors.append((rank, repr(e)))
        finally:
            thread_context_end()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    for c in comms:
        c.close()
    assert not errors, errors
    assert all(o is not None and o[0] == 0 for o in out)
    H = np.concatenate([o[2] for o in out], axis=1)
    fro = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert fro(out[0][1], ref.W) < 1e-8 and fro(H, ref.H) < 1e-8, (fro(out[0][1], ref.W), fro(H, ref.H))
    assert np.array_equal(out[0][1], out[1][1])
