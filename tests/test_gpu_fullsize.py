"""Full-size checks at BASELINE.json's configurations (C2 directly against the oracle; C3 and one
GPU's shard of C4 through size-independent properties).

The update of column j of H depends only on A[:, j], W and W'W, and the update of row i of W only on
A[i, :], H and HH' (plus, for HALS, one global norm per column).  So a full-size device run can be
checked exactly on SAMPLED columns and rows: A is generated on the device by the counter-based
generator, the host regenerates just the sampled columns / rows (same seed, same rounding) and the
oracle solves the small sub-problems.  Every sampled entry exercises the complete contraction over
the other dimension, i.e. the streaming kernel at its full length and split count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def sampled_cols(oracle, m, seed, quant, cols):
    return np.asfortranarray(np.concatenate([oracle.fill_uniform(m, 1, seed, quant=quant, c0=int(c), gheight=m)
                                             for c in cols], axis=1))


def sampled_rows(oracle, m, n, seed, quant, rows):
    return np.asfortranarray(np.concatenate([oracle.fill_uniform(1, n, seed, quant=quant, r0=int(r), gheight=m)
                                             for r in rows], axis=0))


def test_c2_bpp_f32_full_size_against_oracle(gpu):
    """configs[1]: dense 8192 x 4096, k = 16, BPP, A in fp32 -- small enough for the oracle itself."""
    import oracle
    m, n, k = 8192, 4096, 16
    A = oracle.fill_uniform(m, n, 11)
    W0, H0 = oracle.fill_uniform(m, k, 12), oracle.fill_uniform(k, n, 13)
    r = gpu.nmf(A, W0, H0, "BPP", min_iter=1, max_iter=4, tol=1e-12)
    ref = oracle.nmf(A, W0, H0, "BPP", min_iter=1, max_iter=4, tol=1e-12)
    assert r.result == ref.result == 0 and r.iteration_count == ref.iteration_count == 4
    assert np.linalg.norm(r.W - ref.W) / np.linalg.norm(ref.W) < 1e-4     # north_star: 1e-4 relative Frobenius
    assert np.linalg.norm(r.H - ref.H) / np.linalg.norm(ref.H) < 1e-4


def _one_iteration(gpu, m, n, k, alg, storage, seeds):
    from smallk_amd import DenseMatrix, NmfSolver, make_options, uniform_host
    A = DenseMatrix(m, n, storage=storage)
    A.fill_uniform(seeds[0])
    # E[A] = 1/2: scale H0 so that W0 H0 has the same mean (an unscaled uniform start makes the first HALS
    # W update clamp every entry to zero, which would test nothing)
    W0, H0 = uniform_host(m, k, seeds[1]), uniform_host(k, n, seeds[2]) * (2.0 / k)
    s = NmfSolver(A, make_options(m, n, k, alg, normalize=False))
    s.set_factors(W0, H0)
    s.iterate(1)
    assert s.sync() == 0
    W1, H1 = s.factors(normalize=False)
    s.close()
    A.close()
    return W0, H0, W1, H1


def test_c3_hals_bf16_full_size_sampled(gpu):
    """configs[2]: dense 65536 x 16384, k = 32, HALS, A in bf16 (the bench workload)."""
    import oracle
    from oracle import flatclust as of
    m, n, k, seed = 65536, 16384, 32, 101
    W0, H0, W1, H1 = _one_iteration(gpu, m, n, k, "HALS", "bf16", (seed, 102, 103))
    rng = np.random.default_rng(0)
    assert np.all(W1 >= 0) and np.all(H1 >= 0) and np.isfinite(W1).all() and np.isfinite(H1).all()
    # W update (nmf_solver_hals.hpp:66-117): every column is normalised inside the sweep
    assert np.allclose(np.sqrt((W1 * W1).sum(axis=0)), 1.0, rtol=1e-10)
    # sampled rows: t_c[i] = w[i,c] + (R[i,c] - W_cur[i,:] G[:,c]) / G[c,c] must equal nu_c * W1[i,c] with
    # ONE nu_c per column (the column norm the device computed over all 65536 rows)
    rows = np.sort(rng.choice(m, size=40, replace=False))
    Ar = sampled_rows(oracle, m, n, seed, 1, rows)
    G = H0 @ H0.T
    R = Ar @ H0.T
    Wc = W0[rows, :].copy()
    for c in range(k):
        t = Wc[:, c] + (R[:, c] - Wc @ G[:, c]) / G[c, c]
        t[t < 0] = 0.0
        pos = (t > 0) & (W1[rows, c] > 0)
        assert pos.sum() >= 10
        nu = np.median(t[pos] / W1[rows, c][pos])
        assert np.max(np.abs(t - nu * W1[rows, c])) <= 1e-4 * np.max(t), c
        Wc[:, c] = W1[rows, c]                       # Gauss-Seidel: later columns see the normalised value
    # H update with the new W (nmf_solver_hals.hpp:26-62): exact on sampled columns
    cols = np.sort(rng.choice(n, size=48, replace=False))
    Ac = sampled_cols(oracle, m, seed, 1, cols)
    _, _, Hs, _ = of.nnls_hals(Ac, W1, H0[:, cols], 1e-30, 1)     # one sweep, W fixed, no normalisation
    assert relerr(H1[:, cols], Hs) < 1e-4


@pytest.mark.parametrize("alg", ["BPP", "MU"])
def test_c4_shard_f32_sampled(gpu, alg):
    """configs[3] as ONE GPU of the eight sees it: 262144 x 8192 column shard, k = 64, A in fp32.
    MU and BPP update H first, so the oracle on the sub-problem A[:, J] reproduces H1[:, J] exactly;
    the W side is the same statement for the transposed sub-problem A[I, :]' with H1 as the fixed factor."""
    import oracle
    m, n, k, seed = 262144, 8192, 64, 201
    W0, H0, W1, H1 = _one_iteration(gpu, m, n, k, alg, "f32", (seed, 202, 203))
    rng = np.random.default_rng(1)
    cols = np.sort(rng.choice(n, size=k + 8, replace=False))
    Ac = sampled_cols(oracle, m, seed, 0, cols)
    ref = oracle.nmf(Ac, W0, H0[:, cols], alg, min_iter=1, max_iter=1, normalize=False)
    assert ref.result == 0 and relerr(H1[:, cols], ref.H) < 1e-4
    rows = np.sort(rng.choice(m, size=k + 8, replace=False))
    Ar = sampled_rows(oracle, m, n, seed, 0, rows)
    ref = oracle.nmf(np.asfortranarray(Ar.T), np.asfortranarray(H1.T), np.asfortranarray(W0[rows, :].T), alg,
                     min_iter=1, max_iter=1, normalize=False)
    assert ref.result == 0 and relerr(W1[rows, :], ref.H.T) < 1e-4


def test_c4_full_f32_bpp_sampled(gpu):
    """configs[3] WHOLE on one MI355X: dense 262144 x 65536, k = 64, BPP, A in fp32 (A and A' resident:
    137 GB of the 288 GB).  One iteration from a mean-matched start, checked exactly like the shard test:
    the H update on sampled columns and the W update on sampled rows are small NNLS problems the oracle
    solves from regenerated columns / rows of A; every sampled entry went through the full-length
    streaming products (65536- and 262144-long contractions, all row splits) and the k = 64 NNLS kernel."""
    import oracle
    m, n, k, seed = 262144, 65536, 64, 301
    W0, H0, W1, H1 = _one_iteration(gpu, m, n, k, "BPP", "f32", (seed, 302, 303))
    assert np.isfinite(W1).all() and np.isfinite(H1).all() and (W1 >= 0).all() and (H1 >= 0).all()
    rng = np.random.default_rng(2)
    cols = np.sort(rng.choice(n, size=k + 8, replace=False))
    Ac = sampled_cols(oracle, m, seed, 0, cols)
    ref = oracle.nmf(Ac, W0, H0[:, cols], "BPP", min_iter=1, max_iter=1, normalize=False)
    assert ref.result == 0 and relerr(H1[:, cols], ref.H) < 1e-4
    assert np.array_equal(H1[:, cols] > 0, ref.H > 0)                       # same passive sets
    rows = np.sort(rng.choice(m, size=k + 8, replace=False))
    Ar = sampled_rows(oracle, m, n, seed, 0, rows)
    ref = oracle.nmf(np.asfortranarray(Ar.T), np.asfortranarray(H1.T), np.asfortranarray(W0[rows, :].T), "BPP",
                     min_iter=1, max_iter=1, normalize=False)
    assert ref.result == 0 and relerr(W1[rows, :], ref.H.T) < 1e-4


def test_c4_full_size_in_eight_shards_matches_one(gpu):
    """configs[3] in the geometry of the 8-GPU run -- 8 column shards of 8192 columns, 4 row chunks of 8 blocks of 8192
    rows, block-cyclic rows of W, reduce-scatter / all-reduce / all-gather of the packed operand per chunk -- with all 8
    ranks on this box's ONE GPU (one host thread and one device context per rank, the in-process stand-in for RCCL;
    the 137 GB of A and A' are the same bytes as in the one-GPU run, held as 8 shards).  Three BPP iterations; W and the
    columns of H must agree with the unsharded run on the same data to summation order (1e-5), at every rank.  What this
    pins at FULL size is the index arithmetic of the chunk pipeline (64-bit offsets into A', the packed operand and the
    partial products, uneven last blocks are covered by the small tests) -- only the transport differs from 8 GPUs -- and
    the wire type: with fp32 on the wire this comparison reads 1.1e-4 in W after one iteration (sums over 65536 columns
    rounded to 24 bits, amplified by the Gram matrix of noise-like data), with fp64 (the default) 6e-7."""
    import threading
    from smallk_amd import DenseMatrix, NmfSolver, Comm, make_options, uniform_host, thread_context_begin, thread_context_end
    from smallk_amd import dist as sdist
    m, n, k, world, iters = 262144, 65536, 64, 8, 3
    W0 = uniform_host(m, k, 312)
    H0 = uniform_host(k, n, 313) * (2.0 / k)
    opts = dict(min_iter=iters, max_iter=iters, normalize=False)
    # one GPU, unsharded
    A = DenseMatrix(m, n)
    A.fill_uniform(311)
    s = NmfSolver(A, make_options(m, n, k, "BPP", **opts))
    s.set_factors(W0, H0)
    s.iterate(iters)
    assert s.sync() == 0
    W1, H1 = s.factors(normalize=False)
    s.close()
    A.close()
    # eight shards, eight threads
    comms = Comm.init_local(world)
    out, errors = [None] * world, []

    def run(rank):
        try:
            thread_context_begin(0)
            c0, nc = sdist.shard_columns(n, world, rank)
            D = DenseMatrix(m, n, col0=c0, ncols=nc)
            D.fill_uniform(311)                      # the generator is keyed by the global element index
            sv = NmfSolver(D, make_options(m, n, k, "BPP", **opts))
            sv.attach_comm(comms[rank])
            sv.set_factors(W0, H0[:, c0:c0 + nc])
            sv.iterate(iters)
            rc = sv.sync()
            W, H = sv.factors(normalize=False)
            out[rank] = (rc, c0, nc, W if rank in (0, world - 1) else None, H)
            sv.close()
            D.close()
        except Exception as e:          # pragma: no cover
            errors.append((rank, repr(e)))
        finally:
            thread_context_end()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=900)
    for c in comms:
        c.close()
    assert not errors, errors
    assert all(o is not None and o[0] == 0 for o in out)
    H8 = np.concatenate([o[4] for o in out], axis=1)
    fro = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert fro(H8, H1) < 1e-5
    assert fro(out[0][3], W1) < 1e-5 and np.array_equal(out[0][3], out[world - 1][3])      # the gathered W: same bits on every rank
