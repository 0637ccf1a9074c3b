"""Randomised parity sweep of block pivoting at k <= 16 on dense fp32 A -- the path whose NNLS launch packs its own result with
row scales from the a-priori bound x_r <= max|a| / sqrt(G_rr) -- on data chosen to stress that bound and the
fp16 range: well and ill conditioned planted factors, rows / columns scaled over 2^+-12, zero rows and columns, overall scales
2^-30 / 2^+40, starts with negative entries (the first solve must then pack the old way), tolerance-based stopping (snapshot restore).
Every case against the oracle at the 1e-4 bar (the nearly collinear family, cond(W'W) ~ 1e5: as close as the separate launch --
its 1e-4 .. 1e-3 comes from the conditioning and is the same on both paths), with the separate reduce-and-pack launch
(SMK_NNLS_PACK=0) beside it.
(Not swept: data scaled to ~1e-12, where the reference's absolute ZeroizeSmallValues threshold empties the factors of EVERY column
once one column pivots and the run fails as not SPD; the device zeroizes the columns that pivot -- DESIGN 3.)
The fallback message of pack_fail_soft must never appear (grep the stderr of this script).
  python3 tools/fuzz_small_k_bpp.py [cases] [seed] [max_dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, smallk_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
maxdim = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
rng = np.random.default_rng(seed)
smallk_amd.initialize(0)
oracle.set_num_threads(8)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
bad, worst, worst_pair, fams = [], 0.0, 0.0, {}
t0 = time.time()
for case in range(cases):
    k = int(rng.integers(9, 17))
    m = int(rng.integers(4 * k, maxdim)); n = int(rng.integers(4 * k, maxdim))
    fam = str(rng.choice(["planted", "collinear", "colscale", "rowscale", "zeros", "tiny", "huge", "negstart", "noise"]))
    r = k + int(rng.integers(0, 4))
    Wp = rng.random((m, r)) * (rng.random((m, r)) > 0.5)
    Hp = rng.random((r, n)) * (rng.random((r, n)) > 0.5)
    if fam == "collinear":
        Wp = rng.random((m, 1)) + 0.05 * rng.random((m, r))
    A = Wp @ Hp + 0.02 * rng.random((m, n))
    if fam == "noise":
        A = rng.random((m, n))
    if fam == "colscale":
        A = A * np.ldexp(1.0, rng.integers(-12, 13, size=n))[None, :]
    if fam == "rowscale":
        A = A * np.ldexp(1.0, rng.integers(-12, 13, size=m))[:, None]
    if fam == "zeros":
        A[rng.integers(0, m, size=max(1, m // 50)), :] = 0.0
        A[:, rng.integers(0, n, size=max(1, n // 50))] = 0.0
    if fam == "tiny":
        A = np.ldexp(A, -30)
    if fam == "huge":
        A = np.ldexp(A, 40)
    A = oracle.quantize(np.asfortranarray(A), 0)
    W0 = oracle.fill_uniform(m, k, 100 + case)
    H0 = oracle.fill_uniform(k, n, 200 + case) * (2.0 * A.mean() / (0.5 * k))
    if fam == "negstart":
        W0 = W0 - 0.3
    kw = dict(min_iter=1, max_iter=int(rng.integers(2, 25)), tol=1e-14)
    if rng.random() < 0.35:
        kw = dict(min_iter=int(rng.integers(1, 5)), max_iter=int(rng.integers(8, 60)), tol=float(rng.choice([0.1, 0.02, 0.005])), tolcount=int(rng.integers(1, 3)))
    ref = oracle.nmf(A, W0, H0, "BPP", **kw)
    os.environ.pop("SMK_NNLS_PACK", None)
    got = smallk_amd.nmf(A, W0, H0, "BPP", **kw)
    os.environ["SMK_NNLS_PACK"] = "0"
    old = smallk_amd.nmf(A, W0, H0, "BPP", **kw)
    fams[fam] = fams.get(fam, 0) + 1
    desc = f"case {case}: {fam} {m}x{n} k={k} {kw}"
    if got.result != ref.result or (ref.result == 0 and got.iteration_count != ref.iteration_count):
        bad.append(desc + f": result {got.result} vs {ref.result}, iterations {got.iteration_count} vs {ref.iteration_count} (separate launch: {old.result}, {old.iteration_count})")
        continue
    if ref.result != 0:
        continue
    e = max(rel(got.W, ref.W), rel(got.H, ref.H)); p = max(rel(got.W, old.W), rel(got.H, old.H))
    worst = max(worst, e); worst_pair = max(worst_pair, p)
    e_old = max(rel(old.W, ref.W), rel(old.H, ref.H))
    # the nearly collinear family is where the fp16 product form itself leaves the bar (cond(W'W) ~ 1e5; 1e-4 .. 1e-3 on either
    # path): there the packing launch only has to be as close as the separate launch
    if (e > 2 * e_old + 1e-7 if fam == "collinear" else e > 1e-4) or p > 10 * max(e, 1e-7):
        bad.append(desc + f": {e:.2e} from the oracle (the separate launch: {max(rel(old.W, ref.W), rel(old.H, ref.H)):.2e})")
print(f"{cases} cases in {time.time() - t0:.0f} s; families {fams}")
print(f"worst distance to the oracle {worst:.2e}; worst distance between the packing launch and the separate launch {worst_pair:.2e}")
print(f"mismatches: {len(bad)}")
for b in bad:
    print("  " + b)
sys.exit(1 if bad else 0)
