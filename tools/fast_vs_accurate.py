"""profiles/r05_c4_fast_vs_accurate.txt: default product form vs accurate form on the same resident A, five checkpoints per
data set and size (tests/precision_cases.py does the work; tests/test_gpu_precision.py asserts the same numbers).
usage: python tools/fast_vs_accurate.py [c4|mid|all|small|r6] [NSPLIT for the fast leg, e.g. 3]
r6 (round 6, profiles/r06_c3_hals_c4_mu_fast_vs_accurate.txt): C3 under HALS with bf16 A and C4's matrix under MU."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import smallk_amd                                                  # noqa: E402
import oracle                                                      # noqa: E402  (checker only)
from precision_cases import (accurate_iteration_vs_oracle, fast_vs_accurate, hals_accurate_iteration_vs_oracle,   # noqa: E402
                             make_matrix)

which = sys.argv[1] if len(sys.argv) > 1 else "all"
fast_env = sys.argv[2] if len(sys.argv) > 2 else None
smallk_amd.initialize(0)
cases = []
if which in ("c4", "all"):
    cases.append(("C4", 262144, 65536, 64))
if which in ("mid", "all"):
    cases += [("mid", 65536, 16384, 64), ("mid", 32768, 8192, 48)]
if which == "small":
    cases.append(("small", 8192, 4096, 64))
if which == "r6":
    print("# default (fast) product form vs accurate form (SMK_NSPLIT=8), same resident A, same start; bar 1e-4")
    for name, m, n, k, alg, storage in (("C3", 65536, 16384, 32, "HALS", "bf16"), ("C4 matrix", 262144, 65536, 64, "MU", "f32")):
        for data in ("uniform", "planted"):
            t0 = time.time()
            seed = 431 if alg == "HALS" else 441
            A = make_matrix(m, n, data, seed, k, storage=storage)
            if alg == "HALS":
                eW, eH = hals_accurate_iteration_vs_oracle(oracle, A, k, data, seed, k, (seed + 1, seed + 2), quant=1)
                rows, forms, _ = fast_vs_accurate(A, k, alg, (seed + 1, seed + 2), h0_scale=2.0 / k)
            else:
                eH, eW, _ = accurate_iteration_vs_oracle(oracle, A, k, alg, data, seed, k, (seed + 1, seed + 2))
                rows, forms, _ = fast_vs_accurate(A, k, alg, (seed + 1, seed + 2))
            A.close()
            print(f"{name} {m}x{n} k={k} {alg} {storage} {data}: forms fast={forms[0]} accurate={forms[1]}  total {time.time() - t0:.1f}s")
            print(f"   accurate form, 1 iteration vs the oracle / the reference's update formulas on sampled columns and rows: H {eH:.2e}  W {eW:.2e}")
            for it, w, h in rows:
                print(f"   iteration {it:3d}: W {w:.3e}  H {h:.3e}  {'ok' if max(w, h) < 1e-4 else 'OUTSIDE THE BAR'}")
            sys.stdout.flush()
    sys.exit(0)
print("# default (fast) product form vs accurate form (SMK_NSPLIT=8), BPP, fp32 A, same resident A, same start")
print("# relative Frobenius distance of the factors at iterations 1/5/10/25/50; bar 1e-4")
for name, m, n, k in cases:
    for data in ("uniform", "planted"):
        t0 = time.time()
        A = make_matrix(m, n, data, 401, k)
        t1 = time.time()
        eH, eW, same = accurate_iteration_vs_oracle(oracle, A, k, "BPP", data, 401, k, (402, 403))
        rows, forms, _ = fast_vs_accurate(A, k, "BPP", (402, 403), fast_env=fast_env)
        A.close()
        print(f"{name} {m}x{n} k={k} {data}: forms fast={forms[0]} accurate={forms[1]}  generate {t1 - t0:.2f}s  total {time.time() - t0:.1f}s")
        print(f"   accurate form, 1 iteration vs oracle on sampled columns/rows: H {eH:.2e}  W {eW:.2e}  passive sets equal: {same}")
        for it, w, h in rows:
            print(f"   iteration {it:3d}: W {w:.3e}  H {h:.3e}  {'ok' if max(w, h) < 1e-4 else 'OUTSIDE THE BAR'}")
        sys.stdout.flush()
