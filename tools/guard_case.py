"""The run-time guard of the product form on one case: python3 tools/guard_case.py <ill|well> [alg] [k] [iters] [pert]
ill : A = W* H* + noise with nearly collinear columns of W* (cond(W'W) ~ 1e5): the guard must change to the accurate form
well: uniform noise: the guard looks and leaves the fast form alone
Prints one JSON line: product form at the end, guard counters, errors against the oracle."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, smallk_amd as gpu

kind = sys.argv[1] if len(sys.argv) > 1 else "ill"
alg = sys.argv[2] if len(sys.argv) > 2 else "BPP"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 8
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 60
pert = float(sys.argv[5]) if len(sys.argv) > 5 else 0.1          # how far the columns of the planted W* are from collinear
m, n = 3000, 2000
rng = np.random.default_rng(5)
if kind == "ill":
    base = rng.random((m, 1))
    Wp = base + pert * rng.random((m, k))
    Hp = rng.random((k, n)) * (rng.random((k, n)) > 0.5)
    A = Wp @ Hp + 1e-3 * rng.random((m, n))
else:
    A = rng.random((m, n))
A = oracle.quantize(np.asfortranarray(A), 0)
W0 = oracle.fill_uniform(m, k, 43)
H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
gpu.initialize(0)
D = gpu.DenseMatrix.from_host(A)
s = gpu.NmfSolver(D, gpu.make_options(m, n, k, alg, min_iter=iters, max_iter=iters))
form0 = s.product_form()[0]
s.set_factors(W0, H0)
rc, it, _ = s.run()
W, H = s.factors()
form, checks, fired, last = s.product_form()
ref = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print(json.dumps(dict(kind=kind, alg=alg, k=k, rc=rc, iters=it, form_start=form0, form_end=form, guard_checks=checks, guard_fired=fired,
                      cond_times_delta=last, relW=rel(W, ref.W), relH=rel(H, ref.H), ref_rc=ref.result)))
