"""Alternate code paths selected by environment knobs, each in a fresh process (the library reads
them once): streaming-kernel variants (tile shape, ring depth, k-split, loader waves), the
one-launch-per-column HALS W update (fallback for very tall W), operand split counts."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_quick_parity(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "quick_parity.py")], capture_output=True,
                       text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.endswith("OK"), last
    return last


@pytest.mark.parametrize("variant", [0, 1, 5, 6, 7, 9, 11, 13, 14, 15, 18, 20, 21, 108, 110, 111, 115, 125, 126])
def test_streaming_kernel_variants(variant):
    """bf16 A: variants < 100; fp32 A (two fp16 terms, the default): the variants that form is built for"""
    run_quick_parity({"SMK_BP_VARIANT": str(variant)})


@pytest.mark.parametrize("variant", [7, 21, 100, 102, 106, 108, 111, 115, 117])
def test_streaming_kernel_variants_bf16x3(variant):
    """fp32 A as three bf16 terms (SMK_NSPLIT=3): the round-1 kernel (7, 21) and the round-2 kernels"""
    run_quick_parity({"SMK_BP_VARIANT": str(variant), "SMK_NSPLIT": "3"})


@pytest.mark.parametrize("splits", [1, 8])
def test_forced_row_splits(splits):
    run_quick_parity({"SMK_BP_SPLITS": str(splits)})


def test_hals_w_multi_launch_fallback():
    run_quick_parity({"SMK_HALS_W": "multi"})


@pytest.mark.parametrize("env", [{"SMK_NNLS_INV32": "0"}, {"SMK_GRAM_INVERSE_OLD": "1"}, {"SMK_NNLS_INV": "0"}, {"SMK_LD_SKEW": "0"},
                                 {"SMK_NNLS_G16": "0"}, {"SMK_NNLS_G16": "2"}, {"SMK_NNLS_G16_SHAPE": "0"}, {"SMK_NNLS_G16_SHAPE": "1"}, {"SMK_NNLS_G16_SHAPE": "2"}])
def test_older_block_pivoting_routes_stay_selectable(env):
    """k in (16, 32] by masked elimination (the default until late round 4: now through the inverse of the Gram matrix, like
    k in (32, 64]); the 64 x 64 inversion kernel that kept its registers in scratch; no inverse at all; no column-stride skew.
    quick_parity.py has block pivoting at k = 16, 20, 32, 33 and 64."""
    run_quick_parity(env)


def test_two_term_operand_split():
    # 16-bit operand: looser but still inside the bar on these shapes
    run_quick_parity({"SMK_NSPLIT": "2"})


def test_native_fp32_matrix_cores():
    """SMK_NSPLIT=1: v_mfma_f32_32x32x2_f32 on the fp32 tile, no emulation; one iteration on five shapes"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "native_fp32_check.py")], capture_output=True, text=True,
                       env=dict(os.environ, SMK_NSPLIT="1"), cwd=ROOT, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-1500:] + r.stderr[-1500:]


def test_randomised_parity_sweep():
    """tools/fuzz_parity.py: 80 random (shape, rank, algorithm, storage, dense/sparse, stopping rule)
    problems through the C ABI against the oracle.  Longer sweeps (1500 cases, other seeds) were run
    clean while developing; this keeps a slice of it in the suite."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "80", "7"], capture_output=True,
                       text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "all cases within tolerance" in r.stdout


def test_check_every_iteration_loop():
    """SMK_SYNC_PROGRESS=1: the plain check-every-iteration driver loop instead of the one that evaluates
    the stopping rule one iteration late (both must reproduce the oracle's iteration counts: the sweep
    lets the rule fire in ~20 % of its cases)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "60", "5"], capture_output=True,
                       text=True, cwd=ROOT, timeout=600, env=dict(os.environ, SMK_SYNC_PROGRESS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "converged early" in r.stdout


@pytest.mark.parametrize("env", [{"SMK_PROGRESS_DEFER": "0"}, {"SMK_PROGRESS_DEPTH": "3"}, {"SMK_BPP_GRADW": "1"}, {"SMK_HALS_EPILOGUE": "0"},
                                 {"SMK_PROGRESS_DEFER": "0", "SMK_PROGRESS_DEPTH": "2"}, {"SMK_PROGRESS_POLL": "0"},
                                 {"SMK_PROGRESS_TAIL": "0", "SMK_PROGRESS_DEPTH": "2"}, {"SMK_INV_RIDE": "0"}, {"SMK_INV_RIDE": "0", "SMK_INV_STREAM": "1"}, {"SMK_GRAM_RIDE": "0"}])
def test_round_6_switches_stay_selectable(env):
    """The non-default values of round 6's switches on the randomised sweep (tolerance-stopped runs included: iteration counts and
    factors must be the oracle's): the stopping-rule check NOT riding in the next NNLS launch, three checks in flight, BPP's W-side
    gradient formed although it is the NNLS's dual, the HALS sweeps without their packing / Gram epilogues."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "70", "13"], capture_output=True,
                       text=True, cwd=ROOT, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "all cases within tolerance" in r.stdout and "converged early" in r.stdout


def test_progress_check_as_four_stream_operations():
    """SMK_PROGRESS_FUSED=0: gradients, sums, the 64-byte copy and the snapshot as separate stream operations (the path until
    round 6; the default is now ONE launch that also writes the pinned slot, kernels.hip: grad_pg2_fused_kernel).  Same sweep as
    above: iteration counts and factors of tolerance-stopped runs must be the oracle's on both paths."""
    for env in ({"SMK_PROGRESS_FUSED": "0"}, {}):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "60", "5"], capture_output=True,
                           text=True, cwd=ROOT, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
        assert "converged early" in r.stdout


def test_fused_hals_sweep_fails_soft(tmp_path):
    """SMK_HALS_SPIN=1 makes the grid-wide exchange of the fused W sweep give up at once (what would happen if a
    workgroup could not be resident): the run is repeated from the initial factors on the one-launch-per-column
    path, the caller sees OK and the oracle's factors, and the fallback is announced once on stderr."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys; sys.path.insert(0, %r)
import numpy as np, oracle, smallk_amd
smallk_amd.initialize(0)
m, n, k, iters = 3000, 900, 20, 6
A = oracle.fill_uniform(m, n, 42); W0 = oracle.fill_uniform(m, k, 43); H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
ref = oracle.nmf(A, W0, H0, "HALS", min_iter=iters, max_iter=iters)
got = smallk_amd.nmf(A, W0, H0, "HALS", min_iter=iters, max_iter=iters)
e = max(np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W), np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H))
print("RESULT", got.result, got.iteration_count, e)
# the stepwise interface recovers too
D = smallk_amd.DenseMatrix.from_host(A)
s = smallk_amd.NmfSolver(D, smallk_amd.make_options(m, n, k, "HALS", min_iter=iters, max_iter=iters, normalize=False))
s.set_factors(W0, H0); s.iterate(iters); rc = s.sync(); W, H = s.factors()
ref2 = oracle.nmf(A, W0, H0, "HALS", min_iter=iters, max_iter=iters, normalize=False)
print("STEPWISE", rc, np.linalg.norm(W - ref2.W) / np.linalg.norm(ref2.W))
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SMK_HALS_SPIN="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == 0 and int(line[2]) == 6 and float(line[3]) < 1e-4
    step = [l for l in r.stdout.splitlines() if l.startswith("STEPWISE")][0].split()
    assert int(step[1]) == 0 and float(step[2]) < 1e-4
    assert r.stderr.count("repeating the run on the per-column path") == 2


def test_hals_blocked_w_sweep_with_more_rows_than_threads(gpu):
    """HALS above k = 64 sweeps W by blocks of 16 columns with a thread per row (wide.hip: launch_hals_w_update_blocked); the
    grid stops at 1024 workgroups of 256, so beyond 262144 rows a thread takes a second row -- a path no other test reaches.
    Two iterations on 270000 x 96 at k = 70 (five blocks, the last one of six columns) against the oracle."""
    import oracle
    m, n, k = 270000, 96, 70
    rng = np.random.default_rng(3)
    A = (rng.random((m, k)) * (rng.random((m, k)) > 0.7)) @ (rng.random((k, n)) * (rng.random((k, n)) > 0.5)) + 0.05 * rng.random((m, n))
    A = oracle.quantize(A, 0)
    W0, H0 = oracle.fill_uniform(m, k, 5), oracle.fill_uniform(k, n, 6)
    ref = oracle.nmf(A, W0, H0, "HALS", min_iter=2, max_iter=2)
    got = gpu.nmf(A, W0, H0, "HALS", min_iter=2, max_iter=2)
    assert got.result == ref.result == 0 and got.iteration_count == ref.iteration_count
    assert np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W) < 1e-4
    assert np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H) < 1e-4



def test_runtime_guard_of_the_product_form():
    """Opt-in (SMK_GUARD_EVERY=n, BPP): every n iterations the solver compares the fast product form with the accurate one on a
    column sample and changes to the accurate form when cond(Gram) x (product discrepancy) says that one iteration could move
    the factors by the 1e-4 bar (solver.cpp: guard_step).  On data whose planted factor has nearly collinear columns the guard
    must fire at k = 8 -- far below the static k > 64 rule -- and the run must end inside the bar; on uniform noise it must look
    and leave the fast form alone.  Without the switch nothing looks.  (Own processes: the switches are read once per process.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for kind, every in (("ill", "2"), ("well", "4"), ("off", "0")):
        r = subprocess.run([sys.executable, "tools/guard_case.py", "ill" if kind == "off" else kind, "BPP", "8", "40"], cwd=root,
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, SMK_GUARD_EVERY=every))
        assert r.returncode == 0, r.stderr[-2000:]
        out[kind] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ill, well, off = out["ill"], out["well"], out["off"]
    assert ill["form_start"] != 8 and ill["guard_fired"] == 1 and ill["form_end"] == 8, ill
    assert ill["rc"] == ill["ref_rc"] == 0 and ill["relW"] < 1e-4 and ill["relH"] < 1e-4, ill
    assert well["guard_checks"] >= 5 and well["guard_fired"] == 0 and well["form_end"] == well["form_start"], well
    assert well["relW"] < 1e-4 and well["relH"] < 1e-4, well
    assert off["guard_checks"] == 0 and off["form_end"] == off["form_start"], off


def test_small_bpp_takes_the_accurate_form(gpu, monkeypatch):
    """Block pivoting at k in (32, 64] amplifies the product error ~3000 x while the passive sets still move: on data with sparse
    planted factors the fp16 two-term form is 0.4e-4 .. 2e-4 away from the oracle's trajectory around iteration 100 (it contracts
    to 4e-7 by iteration 500; profiles/r04_long_runs_500_iterations.txt).  Where the accurate form costs nothing measurable -- A of
    at most 2^24 entries -- it is the default (the suite otherwise keeps the fp16 form, conftest.py): form 8, and the case that
    peaks at 2e-4 stays at 1e-10 after the same 100 iterations.  Larger matrices, k <= 32 and MU keep the fp16 form."""
    import oracle
    monkeypatch.delenv("SMK_BPP_SMALL_ACCURATE", raising=False)
    m, n, k = 1500, 1100, 64
    rng = np.random.default_rng(17 + k)
    r = k + 2
    A = (rng.random((m, r)) * (rng.random((m, r)) > 0.7)) @ (rng.random((r, n)) * (rng.random((r, n)) > 0.7)) + 0.05 * rng.random((m, n))
    A = oracle.quantize(A, 0)
    W0, H0 = oracle.fill_uniform(m, k, 21), oracle.fill_uniform(k, n, 22)
    D = gpu.DenseMatrix.from_host(A)
    forms = {}
    for alg, kk in (("BPP", 64), ("BPP", 33), ("BPP", 32), ("MU", 64)):
        s = gpu.NmfSolver(D, gpu.make_options(m, n, kk, alg, min_iter=1, max_iter=1))
        forms[(alg, kk)] = s.product_form()[0]
        s.close()
    assert forms == {("BPP", 64): 8, ("BPP", 33): 8, ("BPP", 32): 4, ("MU", 64): 4}, forms
    big = gpu.DenseMatrix(8192, 4096)                     # 2^25 entries: the fp16 form
    big.fill_uniform(1)
    s = gpu.NmfSolver(big, gpu.make_options(8192, 4096, 64, "BPP", min_iter=1, max_iter=1))
    assert s.product_form()[0] == 4
    s.close()
    kw = dict(min_iter=100, max_iter=100, tol=1e-14)
    ref = oracle.nmf(A, W0, H0, "BPP", **kw)
    got = gpu.nmf(A, W0, H0, "BPP", **kw)
    assert got.result == ref.result == 0 and got.iteration_count == ref.iteration_count == 100
    assert np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W) < 1e-9
    assert np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H) < 1e-9
    monkeypatch.setenv("SMK_BPP_SMALL_ACCURATE", "0")     # the fp16 form on the same case: inside 1e-3, not inside the 1e-4 bar
    fast = gpu.nmf(A, W0, H0, "BPP", **kw)
    assert np.linalg.norm(fast.W - ref.W) / np.linalg.norm(ref.W) < 1e-3


@pytest.mark.parametrize("m,n,k", [(2048, 1024, 16), (1999, 1037, 12), (700, 333, 9), (640, 5000, 16), (8192, 4096, 16), (8000, 4100, 13)])
def test_nnls_launch_packs_its_own_result(gpu, monkeypatch, m, n, k):
    """k in (8, 16], BPP, fp32 A, one GPU (C2's shape class): from the second solve of a run on, the NNLS launch writes the packed
    fp16 two-term operand of the factor it solves -- row scales from the a-priori bound x_r <= max|a| / sqrt(G_rr), no pass over the
    result -- and the reduction of its Gram partials rides in the streaming pass that follows (16 extra workgroups).  Same
    iterates as the separate reduce-and-pack launch, inside the bar against the oracle (the two large shapes, C2 among them, fill
    the chip: there the reduction is spread over the first 256 product workgroups, in another summation order); ragged sizes cover the zero padding of the operand, which the last workgroup writes."""
    import oracle
    rng = np.random.default_rng(m + k)
    A = oracle.quantize(np.asfortranarray(rng.random((m, k)) @ rng.random((k, n)) + 0.1 * rng.random((m, n))), 0)
    W0, H0 = oracle.fill_uniform(m, k, 43), oracle.fill_uniform(k, n, 44) * (2.0 / k)
    kw = dict(min_iter=12, max_iter=12, tol=1e-12)
    ref = oracle.nmf(A, W0, H0, "BPP", **kw)
    got = gpu.nmf(A, W0, H0, "BPP", **kw)
    monkeypatch.setenv("SMK_NNLS_PACK", "0")
    old = gpu.nmf(A, W0, H0, "BPP", **kw)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert got.result == old.result == ref.result == 0 and got.iteration_count == ref.iteration_count == 12
    assert rel(got.W, ref.W) < 1e-4 and rel(got.H, ref.H) < 1e-4
    # power-of-two scales are exact and the Gram partials are added in the same order: the two paths agree to the last bit until
    # an entry sits in fp16's subnormal range under one scale and not the other (1e-14), which block pivoting then amplifies like
    # any other rounding (the 8000 x 4100 case: identical for 3 iterations, 2e-14 after 4, 5e-8 after 12, both 2e-6 from the
    # oracle).  That the new path RUNS is shown by the next test.
    assert rel(got.W, old.W) < 1e-6 and rel(got.H, old.H) < 1e-6


def test_nnls_pack_falls_back_when_an_entry_leaves_fp16_range():
    """The launch flags a scaled entry beyond fp16's range (fail flag -4) and smk_solver_run repeats the run from the initial
    factors with the separate pack launch.  The bound cannot fail on valid input; SMK_NNLS_PACK_TEST_ANORM shrinks it 1e6 x.
    Also under SMK_POISON=1 (fresh workspaces filled with NaN bytes): the packing must write every byte the product reads."""
    code = r"""
import sys; sys.path.insert(0, %r)
import numpy as np, oracle, smallk_amd as gpu
gpu.initialize(0)
m, n, k = 1500, 777, 14
rng = np.random.default_rng(3)
A = oracle.quantize(np.asfortranarray(rng.random((m, n))), 0)
W0, H0 = oracle.fill_uniform(m, k, 43), oracle.fill_uniform(k, n, 44)
ref = oracle.nmf(A, W0, H0, "BPP", min_iter=8, max_iter=8)
got = gpu.nmf(A, W0, H0, "BPP", min_iter=8, max_iter=8)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print("RESULT", got.result, got.iteration_count, rel(got.W, ref.W), rel(got.H, ref.H))
""" % ROOT
    for env, expect_msg in (({"SMK_NNLS_PACK_TEST_ANORM": "1e-6"}, True), ({"SMK_POISON": "1"}, False)):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env), cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        res = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
        assert res[1] == "0" and res[2] == "8" and float(res[3]) < 1e-4 and float(res[4]) < 1e-4, res
        assert ("could not pack its result" in r.stderr) == expect_msg, r.stderr[-2000:]


def test_set_factors_uniform_is_the_host_generator(gpu):
    """smk_solver_set_factors_uniform: the start of a run generated on the device must be, bit for bit, the matrices
    smk_uniform_fill_host gives (HierNMF2 draws its initialisers this way since round 4; the oracle replays the host ones)."""
    m, n, k = 700, 333, 5
    A = gpu.DenseMatrix(m, n)
    A.fill_uniform(1)
    s = gpu.NmfSolver(A, gpu.make_options(m, n, k, "MU", min_iter=1, max_iter=1))
    s.set_factors_uniform(12345, 67890)
    W, H = s.factors(normalize=False)
    assert np.array_equal(W, gpu.uniform_host(m, k, 12345)) and np.array_equal(H, gpu.uniform_host(k, n, 67890))
    s2 = gpu.NmfSolver(A, gpu.make_options(m, n, k, "MU", min_iter=3, max_iter=3))
    s2.set_factors(gpu.uniform_host(m, k, 12345), gpu.uniform_host(k, n, 67890))
    s.close()
    s = gpu.NmfSolver(A, gpu.make_options(m, n, k, "MU", min_iter=3, max_iter=3))
    s.set_factors_uniform(12345, 67890)
    assert s.run()[0] == 0 and s2.run()[0] == 0
    Wa, Ha = s.factors()
    Wb, Hb = s2.factors()
    assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)
