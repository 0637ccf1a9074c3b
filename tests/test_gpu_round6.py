"""Round-6 measurement entries of the C ABI on the GPU: smk_solver_iterate_checked (the reference's check-every-iteration loop,
common/include/nmf_solve_generic.hpp:98-121), smk_solver_kernel_name, slot 5 of smk_solver_kernel_time, smk_debug_nnls_stats."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("alg,storage,quant,k", [("BPP", "f32", 0, 16), ("BPP", "f32", 0, 24), ("HALS", "bf16", 1, 32), ("MU", "f32", 0, 8)])
def test_checked_iterations_leave_the_same_factors_and_the_oracles_metric(gpu, alg, storage, quant, k):
    """smk_solver_iterate_checked(n) = smk_solver_iterate(n) plus the stopping rule's metric after every iteration: the factors must
    be bit-identical (forming gradients / snapshots must not touch the state) and the last metric must be the one the oracle's
    NmfSolve<> restatement reports for that iteration (PG_RATIO for HALS / BPP, DELTA_FNORM for MU: smallk.cpp:581-584)."""
    import oracle
    m, n, iters = 900, 700, 7
    A = oracle.fill_uniform(m, n, 71, quant=quant)
    W0 = oracle.fill_uniform(m, k, 72)
    H0 = oracle.fill_uniform(k, n, 73) * (2.0 / k)
    D = gpu.DenseMatrix.from_host(A, storage=storage)
    opts = gpu.make_options(m, n, k, alg, min_iter=iters, max_iter=iters, normalize=False)
    s1, s2 = gpu.NmfSolver(D, opts), gpu.NmfSolver(D, opts)
    for s in (s1, s2):
        s.set_factors(W0, H0)
    s1.iterate(iters)
    assert s1.sync() == 0
    metric = s2.iterate_checked(iters)
    assert s2.sync() == 0
    Wa, Ha = s1.factors(normalize=False)
    Wb, Hb = s2.factors(normalize=False)
    assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)
    # the oracle's metric at the same iteration: a run that checks from iteration 1 on and never stops
    ref = oracle.nmf(A, W0, H0, alg, min_iter=1, max_iter=iters, tol=1e-300, normalize=False)
    assert ref.iteration_count == iters
    want = float(ref.metrics[iters - 1])
    # (the metric is formed from factors that carry the product form's 1e-6 .. 2e-5 distance to the oracle: the parity bar applies)
    assert np.isfinite(want) and abs(metric - want) <= 1e-4 * max(abs(want), 1e-300), (metric, want)
    for s in (s1, s2):
        s.close()
    D.close()


def test_kernel_names_say_which_product_runs(gpu):
    import scipy.sparse as sp
    from smallk_amd import synthetic
    D = gpu.DenseMatrix(2048, 1024, storage="bf16")
    D.fill_uniform(3)
    s = gpu.NmfSolver(D, gpu.make_options(2048, 1024, 16, "HALS"))
    assert s.kernel_name(0).startswith("smk::bigprod_kernel variant") and s.kernel_name(1).startswith("smk::bigprod_kernel variant")
    s.close()
    D.close()
    A = synthetic.term_document(3000, 2000, 60_000, seed=3)                       # ragged columns: entry-balanced segments
    S = gpu.SparseMatrix(A.data, A.indices, A.indptr, A.shape)
    s = gpu.NmfSolver(S, gpu.make_options(3000, 2000, 24, "BPP"))
    assert s.kernel_name(0) == "smk::spmm_seg_kernel" and s.kernel_name(1) == "smk::spmm_seg_kernel"
    s.close()
    S.close()
    Gm = synthetic.community_graph(20000, 16, 8, seed=0)[0]                        # fixed degree: the column-per-lane-group kernel
    S = gpu.SparseMatrix(Gm.data, Gm.indices, Gm.indptr, Gm.shape)
    s = gpu.NmfSolver(S, gpu.make_options(20000, 20000, 24, "BPP"))
    assert s.kernel_name(0) == "smk::spmm_gather_kernel"
    s.close()
    S.close()


def test_nnls_counters_and_timing_slot():
    """SMK_NNLS_STATS=1 (own process: the switch is read once): every column of every solve is counted, the exchange histogram adds up
    to the columns, and slot 5 of smk_solver_kernel_time sees the block-pivoting launches."""
    code = r"""
import sys; sys.path.insert(0, %r)
import ctypes as C, numpy as np, smallk_amd
from smallk_amd import _lib as L
smallk_amd.initialize(0)
m, n, k, iters = 4000, 3000, 24, 4
D = smallk_amd.DenseMatrix(m, n); D.fill_uniform(5)
s = smallk_amd.NmfSolver(D, smallk_amd.make_options(m, n, k, "BPP", min_iter=iters, max_iter=iters))
s.set_factors_uniform(6, 7)
s.iterate(0); s.sync()
out = (C.c_uint64 * 256)()
assert L.lib().smk_debug_nnls_stats(out, 1) == 0
s.enable_timing(True)
import os
s.iterate(iters); assert s.sync() == 0
assert L.lib().smk_debug_nnls_stats(out, 0) == 0
c = np.array(out[:], dtype=np.int64)
assert c[178] == iters * (m + n), c[178]
assert c[0:16].sum() == c[178]
assert c[16:81].sum() == c[178]                     # one first solve per column
ms, cnt = s.kernel_time(5)
assert cnt >= 1 and ms > 0.0, (ms, cnt)
print("stats OK", int(c[178]), round(ms, 3), cnt)
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=600,
                       env=dict(os.environ, SMK_NNLS_STATS="1", SMK_TIMING_STRIDE="1"))
    assert r.returncode == 0 and "stats OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_check_totals_ride_in_the_pass_tail():
    """C2's shape (8192 x 4096, k = 16, BPP, fp32), tolerance-stopped: by default the stopping-rule check of iteration i is formed
    inside iteration i + 1 -- the per-workgroup sums by the H-side NNLS launch, the totals by one more tail workgroup of the pass
    behind it (bigprod.hip: check_totals_tail), and the host polls the pinned slot for the tag stored behind the result instead of
    waiting for an event.  SMK_PROGRESS_TAIL=0 (totals as a launch of their own), SMK_PROGRESS_POLL=0 (an event again),
    SMK_PROGRESS_DEFER=0 (the whole check as launches behind the iteration) and SMK_SYNC_PROGRESS=1 (no speculation at all) must
    stop at the same iteration with bit-identical factors (nmf_solve_generic.hpp:98-121)."""
    code = r"""
import sys, os, hashlib; sys.path.insert(0, %r)
import numpy as np, smallk_amd as g
g.initialize(0)
m, n, k = 8192, 4096, 16
D = g.DenseMatrix(m, n, storage="f32"); D.fill_planted(5, k, 0.7, 0.05)
s = g.NmfSolver(D, g.make_options(m, n, k, "BPP", min_iter=3, max_iter=200, tol=float(sys.argv[1]), normalize=False))
s.set_factors_uniform(6, 7)
rc, iters, _ = s.run()
W, H = s.factors(normalize=False)
print("RESULT", rc, iters, hashlib.sha1(W.tobytes()).hexdigest(), hashlib.sha1(H.tobytes()).hexdigest(), "|", s.kernel_name(2))
""" % ROOT
    legs = [({}, "tail of the pass"), ({"SMK_PROGRESS_TAIL": "0"}, "totals as one launch"),
            ({"SMK_PROGRESS_POLL": "0"}, "tail of the pass"), ({"SMK_PROGRESS_TAIL": "0", "SMK_PROGRESS_POLL": "0"}, "totals as one launch"),
            ({"SMK_PROGRESS_DEFER": "0"}, "launches of its own"), ({"SMK_PROGRESS_DEFER": "0", "SMK_PROGRESS_POLL": "0"}, "launches of its own"),
            ({"SMK_SYNC_PROGRESS": "1"}, "launches of its own")]
    seen = []
    for env, route in legs:
        r = subprocess.run([sys.executable, "-c", code, "0.005"], capture_output=True, text=True, cwd=ROOT, timeout=600,
                           env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert r.returncode == 0 and line, r.stdout[-1500:] + r.stderr[-1500:]
        head, how = line[0].split("|")
        assert route in how, (env, how)
        for other in ("tail of the pass", "totals as one launch"):
            assert other == route or other not in how, (env, how)
        seen.append(head.split()[1:])
    assert int(seen[0][0]) == 0 and 3 < int(seen[0][1]) < 200, seen[0]          # the rule fired, not the iteration limit
    assert all(x == seen[0] for x in seen), seen


@pytest.mark.parametrize("shape", ["ragged", "fixed_degree", "dense", "dense_bf16"])
def test_gram_inverse_rides_in_the_product_launch(shape):
    """BPP at k in (16, 64]: the Gram inverse the next block-pivoting launch needs is formed by one more workgroup of the product
    launch that follows the Gram matrix (InvRide; gram_inverse.h) -- the gather product on ragged columns (spmm_seg_kernel) and on a
    fixed-degree graph (spmm_gather_kernel), the streaming product of a dense matrix (bigprod_f3_kernel for fp32 storage, bigprod_kernel for bf16, bigprod_f64_kernel in
    the accurate form that small problems take).  In stream order
    (SMK_INV_RIDE=0) and beside the product on a second stream (+ SMK_INV_STREAM=1, the route until round 6) the same elimination
    runs as a launch of its own: bit-identical factors, and the oracle's to the parity bar (nnls.hpp:144-244,
    nmf_solver_bpp.hpp:342-377)."""
    code = r"""
import sys, hashlib; sys.path.insert(0, %r)
import numpy as np, scipy.sparse as sp, oracle, smallk_amd as g
g.initialize(0)
rng = np.random.default_rng(11)
shape, out = sys.argv[1], []
for k in (24, 32, 48, 64):
    m, n = 1500, 1100
    if shape.startswith("dense"):
        m, n = 4096, 1536
        A = oracle.fill_uniform(m, n, 5 + k, quant=1 if shape == "dense_bf16" else 0)
        W0, H0 = oracle.fill_uniform(m, k, 3), oracle.fill_uniform(k, n, 4)
        got = g.nmf(A, W0, H0, "BPP", min_iter=5, max_iter=5, storage="bf16" if shape == "dense_bf16" else "f32")
        ref = oracle.nmf(A, W0, H0, "BPP", min_iter=5, max_iter=5)
        err = max(np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W), np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H))
        assert err < 1e-4, (k, err)
        out.append(hashlib.sha1(got.W.tobytes() + got.H.tobytes()).hexdigest())
        continue
    if shape == "ragged":
        A = sp.random(m, n, density=0.02, random_state=5, format="csc")
    else:                                   # 12 stored entries in every column
        rows = np.concatenate([rng.choice(m, size=12, replace=False) for _ in range(n)])
        A = sp.csc_matrix((rng.random(12 * n) + 0.1, rows, np.arange(0, 12 * n + 1, 12)), shape=(m, n))
        A.sort_indices()
    W0, H0 = oracle.fill_uniform(m, k, 3), oracle.fill_uniform(k, n, 4)
    got = g.nmf_sparse(A, W0, H0, "BPP", min_iter=6, max_iter=6)
    ref = oracle.nmf(np.asfortranarray(A.toarray()), W0, H0, "BPP", min_iter=6, max_iter=6)
    err = max(np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W), np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H))
    assert err < 1e-8, (k, err)
    out.append(hashlib.sha1(got.W.tobytes() + got.H.tobytes()).hexdigest())
print("RESULT", " ".join(out))
""" % ROOT
    seen = []
    for env in ({}, {"SMK_INV_RIDE": "0"}, {"SMK_INV_RIDE": "0", "SMK_INV_STREAM": "1"}, {"SMK_INV_RIDE": "0", "SMK_INV_STREAM": "0"}):
        r = subprocess.run([sys.executable, "-c", code, shape], capture_output=True, text=True, cwd=ROOT, timeout=600, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert r.returncode == 0 and line, r.stdout[-1500:] + r.stderr[-1500:]
        seen.append(line[0])
    assert all(x == seen[0] for x in seen), seen


def test_gram_matrix_rides_in_the_sparse_product_launches():
    """Sparse A, k in (8, 32] (MU, HALS, BPP at k <= 16): the Gram matrix of a factor is formed inside the two launches of the gather
    product that follows it in every schedule -- partial sums by extra workgroups of spmm_seg_kernel, their reduction by extra
    workgroups of the fix-up launch (gram_body.h, common.h: GramRide).  The same partial sums in the same order as the launches of
    their own (SMK_GRAM_RIDE=0): bit-identical factors, and the oracle's (nmf_solver_hals.hpp:166-199, nmf_solver_mu.hpp:121-164).
    One matrix has long columns (the reduction rides in the fix-up launch), one has none (it goes out as a launch of its own)."""
    code = r"""
import sys, hashlib; sys.path.insert(0, %r)
import numpy as np, scipy.sparse as sp, oracle, smallk_amd as g
g.initialize(0)
out = []
m, n = 1500, 1100
mats = [sp.random(m, n, density=0.15, random_state=5, format="csc"),            # columns of ~225 entries: long ones among them
        sp.random(m, n, density=0.02, random_state=6, format="csc")]            # ~30 entries per column: no fix-up launch
for ai, A in enumerate(mats):
    Ad = np.asfortranarray(A.toarray())
    for alg, ks in (("HALS", (12, 16, 24, 32)), ("MU", (16, 32)), ("BPP", (12, 16))):
        for k in ks:
            if alg == "HALS" and ai == 1 and k > 12: continue                   # (rows of H collapse there: the run sits on the epsilon guard)
            W0, H0 = oracle.fill_uniform(m, k, 3), oracle.fill_uniform(k, n, 4) * (2.0 / k)
            got = g.nmf_sparse(A, W0, H0, alg, min_iter=6, max_iter=6)
            ref = oracle.nmf(Ad, W0, H0, alg, min_iter=6, max_iter=6)
            err = max(np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W), np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H))
            assert err < 1e-8, (ai, alg, k, err)
            out.append(hashlib.sha1(got.W.tobytes() + got.H.tobytes()).hexdigest()[:12])
# a matrix without stored entries: no gather launch exists to ride in -- the Gram matrix is formed by itself, as on the other route
E = sp.csc_matrix((200, 150))
W0, H0 = oracle.fill_uniform(200, 12, 3), oracle.fill_uniform(12, 150, 4)
got = g.nmf_sparse(E, W0, H0, "MU", min_iter=3, max_iter=3)
out.append(hashlib.sha1(got.W.tobytes() + got.H.tobytes()).hexdigest()[:12])
print("RESULT", " ".join(out))
""" % ROOT
    seen = []
    for env in ({}, {"SMK_GRAM_RIDE": "0"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=900, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert r.returncode == 0 and line, r.stdout[-1500:] + r.stderr[-1500:]
        seen.append(line[0])
    assert seen[0] == seen[1], seen


@pytest.mark.parametrize("flags", [["--check-every-iteration"], ["--api-path"]])
def test_bench_flags_of_round_6_run(flags):
    """bench.py --check-every-iteration / --api-path on the smallest workload (C1: 512 x 256, k = 8, MU): one JSON line with the
    contract's keys, the flag's own fields, and a rate that is a rate."""
    import json
    r = subprocess.run([sys.executable, "bench.py", "--workload", "c1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"] + flags,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-1500:]
    out = json.loads(line[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in out, key
    assert out["value"] > 0 and out["n_gpus"] == 1 and out["steps"] == 5
    if "--check-every-iteration" in flags:
        assert out["config"]["progress_checks"].startswith("after EVERY timed iteration")
        assert out["config"]["check_route"] and "none formed" not in out["config"]["check_route"]
    else:
        assert out["config"]["progress_checks"].startswith("none in the timed region")
        ap = out["api_path"]
        assert ap["iterations"] == 7 and ap["end_to_end_it_s"] > 0 and ap["upload_GBps"] > 0 and ap["m"] == 512 and ap["n"] == 256
        assert ap["one_call_s"] >= ap["solver_elapsed_s"] * 0.5
