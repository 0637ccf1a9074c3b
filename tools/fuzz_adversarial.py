"""Randomised parity sweep on data chosen to stress scales, thresholds and degenerate structure, every algorithm and rank class,
dense (fp32 / bf16 storage) and sparse: rows / columns scaled over 2^+-12, zero rows and columns, overall scales 2^-20 .. 2^40,
duplicated columns, a few huge outliers, constant matrices, starts with zero entries.  Against the oracle at the 1e-4 bar
(1e-8 sparse); result codes and iteration counts must agree.
  python3 tools/fuzz_adversarial.py [cases] [seed] [max_dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import oracle, smallk_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
maxdim = int(sys.argv[3]) if len(sys.argv) > 3 else 1400
rng = np.random.default_rng(seed)
smallk_amd.initialize(0)
oracle.set_num_threads(8)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
bad, aside, fams, worst = [], [], {}, {}
t0 = time.time()
for case in range(cases):
    alg = str(rng.choice(["MU", "HALS", "BPP", "RANK2"]))
    k = 2 if alg == "RANK2" else int(rng.choice([rng.integers(2, 9), rng.integers(9, 17), rng.integers(17, 33), rng.integers(33, 65), rng.integers(65, 131)]))
    m = int(rng.integers(4 * k + 8, maxdim)); n = int(rng.integers(4 * k + 8, maxdim))
    sparse = rng.random() < 0.25
    storage = "f32" if sparse else str(rng.choice(["f32", "bf16"]))
    fam = str(rng.choice(["colscale", "rowscale", "zeros", "small", "huge", "dupcols", "outliers", "constant", "zerostart"]))
    r = k + int(rng.integers(1, 4))
    A = (rng.random((m, r)) * (rng.random((m, r)) > 0.5)) @ (rng.random((r, n)) * (rng.random((r, n)) > 0.5)) + 0.02 * rng.random((m, n))
    if fam == "colscale": A = A * np.ldexp(1.0, rng.integers(-12, 13, size=n))[None, :]
    if fam == "rowscale": A = A * np.ldexp(1.0, rng.integers(-12, 13, size=m))[:, None]
    if fam == "zeros":
        A[rng.integers(0, m, size=max(1, m // 40)), :] = 0.0
        if alg not in ("BPP",) or not sparse: A[:, rng.integers(0, n, size=max(1, n // 40))] = 0.0
    if fam == "small": A = np.ldexp(A, -20)
    if fam == "huge": A = np.ldexp(A, 40)
    if fam == "dupcols":
        src = rng.integers(0, n, size=n // 4); dst = rng.integers(0, n, size=n // 4)
        A[:, dst] = A[:, src]
    if fam == "outliers":
        A[rng.integers(0, m, size=5), rng.integers(0, n, size=5)] *= 1.0e4
    if fam == "constant": A = np.full((m, n), 0.75) + 1e-3 * rng.random((m, n))
    if sparse:
        A = A * (rng.random((m, n)) < 0.15)
        A[:, A.sum(axis=0) == 0] += 1e-3 * A.max()
        As = sp.csc_matrix(A)
    quant = 1 if storage == "bf16" else 0
    Aq = A if sparse else oracle.quantize(np.asfortranarray(A), quant)
    W0 = oracle.fill_uniform(m, k, 100 + case)
    H0 = oracle.fill_uniform(k, n, 200 + case) * (2.0 * max(A.mean(), 1e-300) / (0.5 * k))
    if fam == "zerostart":
        W0[rng.random((m, k)) < 0.2] = 0.0
        H0[rng.random((k, n)) < 0.2] = 0.0
    kw = dict(min_iter=1, max_iter=int(rng.integers(2, 12)), tol=1e-14)
    if rng.random() < 0.3:
        kw = dict(min_iter=int(rng.integers(1, 5)), max_iter=int(rng.integers(6, 40)), tol=float(rng.choice([0.1, 0.02, 0.005])), tolcount=int(rng.integers(1, 3)),
                  prog_est=int(rng.integers(0, 2)))
    if os.environ.get("SMK_FUZZ_ONLY") and case != int(os.environ["SMK_FUZZ_ONLY"]):      # replay one case (the draws above keep the stream)
        continue
    ref = oracle.nmf_sparse(As, W0, H0, alg, **kw) if sparse else oracle.nmf(Aq, W0, H0, alg, **kw)
    got = smallk_amd.nmf_sparse(As, W0, H0, alg, **kw) if sparse else smallk_amd.nmf(A, W0, H0, alg, storage=storage, **kw)
    if os.environ.get("SMK_FUZZ_ONLY"):
        print(f"case {case}: ref result {ref.result} after {ref.iteration_count}; device result {got.result} after {got.iteration_count}; dead rows of the device H: "
              f"{int((np.abs(got.H).max(axis=1) == 0).sum())}, zero columns of the device W: {int((np.abs(got.W).max(axis=0) == 0).sum())}")
    fams[fam] = fams.get(fam, 0) + 1
    desc = f"case {case}: {alg} {fam} {m}x{n} k={k} {'sparse' if sparse else storage} {kw}"
    if got.result != ref.result or (ref.result == 0 and got.iteration_count != ref.iteration_count):
        if got.result == ref.result == 0 and "tolcount" in kw:        # a stopping rule that fired elsewhere: judge the trajectory at the earlier stop
            it = min(got.iteration_count, ref.iteration_count)
            kw = dict(min_iter=it, max_iter=it, tol=1e-14)
            ref = oracle.nmf_sparse(As, W0, H0, alg, **kw) if sparse else oracle.nmf(Aq, W0, H0, alg, **kw)
            got = smallk_amd.nmf_sparse(As, W0, H0, alg, **kw) if sparse else smallk_amd.nmf(A, W0, H0, alg, storage=storage, **kw)
            desc += f" [stopped at {got.iteration_count} vs {ref.iteration_count}; compared after {it}]"
        elif alg == "BPP" and k > 64 and ref.result != 0 and got.result == 0:
            # (3) above k = 64 the device eliminates in another order than dpotrf (inverse / tile kernels): on a
            #     passive block that is singular to rounding the reference meets a pivot <= 0 and stops, the device a tiny positive one
            aside.append(desc + f": the reference stops as not SPD at iteration {ref.iteration_count + 1}, the device goes on (numerically singular block, k > 64)")
            continue
        else:
            bad.append(desc + f": result {got.result} vs {ref.result}, iterations {got.iteration_count} vs {ref.iteration_count}")
            continue
    if ref.result != 0:
        fams["fails in both"] = fams.get("fails in both", 0) + 1
        continue
    e = max(rel(got.W, ref.W), rel(got.H, ref.H))
    key = (alg, "sparse" if sparse else storage)
    if not (e <= (1e-8 if sparse else 1e-4)):
        # Two kinds of case have no stable answer to compare and are set aside (counted, listed):
        # (1) HALS with two or more components dead at once: the reference resets their W columns to the same constant vector
        #     (nmf_solver_hals.hpp:105-111), after which the H update of the second one is rounding noise (entries of 9e-16)
        #     that the next W update divides by its own square -- the reference's own result depends on summation order;
        # (2) Gram matrices conditioned so badly that the reduced-precision product decides the iterate: the accurate product
        #     form (SMK_NSPLIT=8) must then agree with the oracle.
        why = None
        if alg == "HALS":
            for it in range(1, min(ref.iteration_count, 12) + 1):
                kw1 = dict(kw, min_iter=it, max_iter=it, tol=1e-14, normalize=False)
                r1 = oracle.nmf_sparse(As, W0, H0, alg, **kw1) if sparse else oracle.nmf(Aq, W0, H0, alg, **kw1)
                dead = int((np.abs(r1.H).max(axis=1) < 1e-12 * max(np.abs(r1.H).max(), 1e-300)).sum())
                if dead >= 2:
                    why = f"degenerate in the reference: {dead} components dead after iteration {it}"
                    break
        if why is None and not sparse:
            os.environ["SMK_NSPLIT"] = "8"
            acc = smallk_amd.nmf(A, W0, H0, alg, storage=storage, **kw)
            del os.environ["SMK_NSPLIT"]
            ea = max(rel(acc.W, ref.W), rel(acc.H, ref.H)) if acc.result == 0 and acc.iteration_count == ref.iteration_count else float("inf")
            if ea <= 1e-6:
                why = f"ill-conditioned: the accurate product form is {ea:.1e} from the oracle"
        if why:
            aside.append(desc + f": {e:.2e} from the oracle -- {why}")
            continue
        bad.append(desc + f": {e:.2e} from the oracle")
        continue
    worst[key] = max(worst.get(key, 0.0), e)
print(f"{cases} cases in {time.time() - t0:.0f} s; families {fams}")
print("worst distance to the oracle: " + ", ".join(f"{a} {s} {v:.1e}" for (a, s), v in sorted(worst.items())))
print(f"set aside (no stable answer to compare): {len(aside)}")
for b in aside:
    print("  " + b)
print(f"mismatches: {len(bad)}")
for b in bad:
    print("  " + b)
sys.exit(1 if bad else 0)
