"""Single-copy matrices (no stored transpose): MU and HALS with the H*A' pass taken from A itself (bigprod.hip, transposed
source) -- the reference's MU / HALS call Gemm(NORMAL, TRANSPOSE) on A (nmf_solver_mu.hpp:121-164, nmf_solver_hals.hpp:166-199).
Same results as with the stored transpose (the same products, the usual bars against the oracle), half the footprint; a consumer of
the transpose (BPP, RANK2, the accurate form) makes the matrix build it on demand."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(gpu, A, W0, H0, alg, iters, single, storage="bf16"):
    from smallk_amd import DenseMatrix, NmfSolver, make_options
    D = DenseMatrix.from_host(A, storage=storage, single_copy=single)
    assert D.single_copy == single
    s = NmfSolver(D, make_options(A.shape[0], A.shape[1], W0.shape[1], alg, normalize=False))
    s.set_factors(W0, H0)
    s.iterate(iters)
    assert s.sync() == 0
    W, H = s.factors(normalize=False)
    nbytes = D.device_bytes
    s.close()
    D.close()
    return W, H, nbytes


@pytest.mark.parametrize("storage,quant", [("bf16", 1), ("f32", 0)])
@pytest.mark.parametrize("alg", ["MU", "HALS", "BPP"])
@pytest.mark.parametrize("m,n,k", [(512, 256, 8), (700, 1100, 17), (1000, 333, 32), (2051, 1500, 33), (640, 4100, 64), (900, 800, 100),
                                   (300, 17000, 16)])
def test_single_copy_equals_stored_transpose_and_oracle(gpu, alg, m, n, k, storage, quant):
    """bf16: the transposing LDS read; fp32: eight strided 4-byte reads per operand, fp16 two-term form (MU) / bf16x3 (HALS); the last
    shape has a contraction long enough for the 4-stage fold interval."""
    import oracle
    if alg in ("HALS", "BPP") and k > 64:
        pytest.skip("HALS and BPP above k = 64 take the accurate form, which builds the stored transpose (tested below)")
    A = oracle.fill_uniform(m, n, 7, quant=quant)
    W0 = oracle.fill_uniform(m, k, 8)
    H0 = oracle.fill_uniform(k, n, 9) * (2.0 / k)
    iters = 6
    W1, H1, b1 = _run(gpu, A, W0, H0, alg, iters, single=True, storage=storage)
    W2, H2, b2 = _run(gpu, A, W0, H0, alg, iters, single=False, storage=storage)
    fro = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    # the same products with other fp32 accumulation chains (fold intervals, kernel shapes); HALS amplifies product-level
    # differences several thousand times (solver.cpp): measured 2e-6 .. 3.3e-6 on fp32 A, below 1e-6 everywhere else
    bar = 2e-5 if alg in ("HALS", "BPP") else 1e-6
    assert fro(W1, W2) < bar and fro(H1, H2) < bar, (fro(W1, W2), fro(H1, H2))
    ref = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, normalize=False)
    assert fro(W1, ref.W) < 1e-4 and fro(H1, ref.H) < 1e-4, (fro(W1, ref.W), fro(H1, ref.H))
    assert b1 < 0.62 * b2                       # the footprint the ABI reports: no second copy


def test_single_copy_builds_the_transpose_when_something_needs_it(gpu):
    """RANK2 and the accurate form read the stored transpose: the first such solver on a single-copy matrix allocates and
    fills it (the matrix is an ordinary one afterwards); MU / HALS solvers created before keep reading A."""
    import oracle
    from smallk_amd import DenseMatrix, NmfSolver, make_options
    m, n, k = 600, 400, 8
    A = oracle.fill_uniform(m, n, 3, quant=1)
    W0, H0 = oracle.fill_uniform(m, k, 4), oracle.fill_uniform(k, n, 5) * (2.0 / k)
    D = DenseMatrix.from_host(A, storage="bf16", single_copy=True)
    b0 = D.device_bytes
    mu = NmfSolver(D, make_options(m, n, k, "MU", normalize=False))       # planned on the transposed source
    mu.set_factors(W0, H0)
    assert D.single_copy
    os.environ["SMK_NSPLIT"] = "8"
    try:
        bpp = NmfSolver(D, make_options(m, n, k, "BPP", normalize=False)) # the accurate form needs A': built now
    finally:
        del os.environ["SMK_NSPLIT"]
    assert not D.single_copy and D.device_bytes > 1.6 * b0
    bpp.set_factors(W0, H0)
    for s, alg in ((mu, "MU"), (bpp, "BPP")):
        s.iterate(4)
        assert s.sync() == 0
        W, H = s.factors(normalize=False)
        ref = oracle.nmf(A, W0, H0, alg, min_iter=4, max_iter=4, normalize=False)
        assert np.linalg.norm(W - ref.W) / np.linalg.norm(ref.W) < 1e-4 and np.linalg.norm(H - ref.H) / np.linalg.norm(ref.H) < 1e-4
        s.close()
    D.close()


def test_single_copy_full_size_c3(gpu):
    """configs[2] (65536 x 16384, k = 32, HALS, bf16) on a single-copy matrix: three iterations against the stored-transpose run
    on the same generated data."""
    from smallk_amd import DenseMatrix, NmfSolver, make_options, uniform_host
    m, n, k = 65536, 16384, 32
    W0, H0 = uniform_host(m, k, 102), uniform_host(k, n, 103) * (2.0 / k)
    out = []
    for single in (True, False):
        D = DenseMatrix(m, n, storage="bf16", single_copy=single)
        D.fill_uniform(101)
        s = NmfSolver(D, make_options(m, n, k, "HALS", normalize=False))
        s.set_factors(W0, H0)
        s.iterate(3)
        assert s.sync() == 0
        out.append(s.factors(normalize=False) + (D.device_bytes,))
        s.close()
        D.close()
    fro = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert fro(out[0][0], out[1][0]) < 1e-6 and fro(out[0][1], out[1][1]) < 1e-6
    assert out[0][2] * 2 <= out[1][2] * 1.01


def test_matrix_beyond_half_of_hbm_falls_back_to_a_single_copy(gpu):
    """262144 x 196608 fp32 = 206 GB: A fits the 288 GB of one MI355X, A and A' together do not.  smk_matrix_create then makes the
    matrix a single copy by itself; one BPP iteration at k = 64 (the fp16 two-term form, both products from A) is checked on sampled
    columns and rows against the oracle exactly as the C4 test does -- 1.5 x C4's matrix on one GPU."""
    import oracle
    from smallk_amd import DenseMatrix, NmfSolver, make_options, uniform_host
    import gc
    from smallk_amd import trim_device_cache
    m, n, k, seed = 262144, 196608, 64, 601
    gc.collect()                                           # matrices of earlier tests that were left to the collector
    trim_device_cache()
    A = DenseMatrix(m, n)                                  # asks for both copies
    try:
        assert A.single_copy and A.device_bytes < 215e9
        A.fill_uniform(seed)
        W0, H0 = uniform_host(m, k, 602), uniform_host(k, n, 603) * (2.0 / k)
        s = NmfSolver(A, make_options(m, n, k, "BPP", normalize=False))
        assert s.product_form()[0] == 4
        s.set_factors(W0, H0)
        s.iterate(1)
        assert s.sync() == 0
        W1, H1 = s.factors(normalize=False)
        s.close()
        assert A.single_copy
    finally:
        A.close()
    relerr = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
    rng = np.random.default_rng(6)
    cols = np.sort(rng.choice(n, size=k + 8, replace=False))
    Ac = np.asfortranarray(np.concatenate([oracle.fill_uniform(m, 1, seed, c0=int(c), gheight=m) for c in cols], axis=1))
    ref = oracle.nmf(Ac, W0, H0[:, cols], "BPP", min_iter=1, max_iter=1, normalize=False)
    assert ref.result == 0 and relerr(H1[:, cols], ref.H) < 1e-4
    rows = np.sort(rng.choice(m, size=k + 8, replace=False))
    Ar = np.asfortranarray(np.concatenate([oracle.fill_uniform(1, n, seed, r0=int(r), gheight=m) for r in rows], axis=0))
    ref = oracle.nmf(np.asfortranarray(Ar.T), np.asfortranarray(H1.T), np.asfortranarray(W0[rows, :].T), "BPP",
                     min_iter=1, max_iter=1, normalize=False)
    assert ref.result == 0 and relerr(W1[rows, :], ref.H.T) < 1e-4


@pytest.mark.parametrize("alg,storage,quant", [("BPP", "f32", 0), ("MU", "f32", 0), ("HALS", "bf16", 1), ("BPP", "bf16", 1)])
def test_column_sharded_run_on_single_copy_shards(gpu, alg, storage, quant):
    """Three column shards through the in-process stand-in communicator, every shard a single copy: the chunked H*A' pass
    (reduce-scatter / all-reduce per row chunk) reads row ranges of A itself.  Against the oracle on the whole matrix."""
    import threading
    import oracle
    from smallk_amd import Comm, DenseMatrix, NmfSolver, make_options, thread_context_begin, thread_context_end
    from smallk_amd import dist as sdist
    m, n, k, iters, world = 20000, 3000, 24, 4, 3
    A = oracle.fill_uniform(m, n, 31, quant=quant)
    W0 = oracle.fill_uniform(m, k, 32)
    H0 = oracle.fill_uniform(k, n, 33) * (2.0 / k)
    ref = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, normalize=False)
    comms = Comm.init_local(world)
    out, errors = [None] * world, []

    def run(rank):
        try:
            thread_context_begin(0)
            c0, nc = sdist.shard_columns(n, world, rank)
            D = DenseMatrix(m, n, col0=c0, ncols=nc, storage=storage, single_copy=True)
            D.upload(A[:, c0:c0 + nc])
            sv = NmfSolver(D, make_options(m, n, k, alg, min_iter=iters, max_iter=iters, normalize=False))
            sv.attach_comm(comms[rank])
            sv.set_factors(W0, H0[:, c0:c0 + nc])
            sv.iterate(iters)
            rc = sv.sync()
            W, H = sv.factors(normalize=False)
            out[rank] = (rc, W, H, D.single_copy)
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
        t.join(timeout=300)
    for c in comms:
        c.close()
    assert not errors, errors
    assert all(o is not None and o[0] == 0 and o[3] for o in out)          # still single copies after the run
    H = np.concatenate([o[2] for o in out], axis=1)
    fro = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert fro(out[0][1], ref.W) < 1e-4 and fro(H, ref.H) < 1e-4, (fro(out[0][1], ref.W), fro(H, ref.H))
    assert np.array_equal(out[0][1], out[world - 1][1])


@pytest.mark.parametrize("mode,env", [("guard", {"SMK_GUARD_EVERY": "1", "SMK_GUARD_TAU": "1e-30"}), ("nnls_hals", {})])
def test_single_copy_that_switches_to_the_accurate_form_in_mid_run(mode, env):
    """every re-plan (the run-time guard of the product form, NnlsHals' switch to the accurate form, the form agreement of a
    sharded run) used to keep the transposed-source plan of a single-copy matrix while the solver handed the fp64 factor to it --
    garbage without an error.  plan_products now builds the stored transpose first.  1300 rows: 0 < m mod 256 <= 128, the shape whose
    last tile used to read past the padded rows of A."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "single_copy_replan.py"), mode], capture_output=True, text=True,
                       env=dict(os.environ, **env), cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["single_at_start"] and j["form_start"] != 8 and j["form_end"] == 8 and not j["single_at_end"], j
    if mode == "guard":
        assert j["rc"] == 0 and j["guard_fired"] >= 1 and j["relW"] < 1e-4 and j["relH"] < 1e-4, j
    else:
        assert j["rc"] == 0 and j["iterations"] == j["ref_iterations"] and j["relH"] < 1e-6, j
