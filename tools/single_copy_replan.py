"""A single-copy matrix whose run LEAVES what the transposed-source kernels cover after the solver was created:
   python3 tools/single_copy_replan.py guard      BPP with SMK_GUARD_EVERY=1, SMK_GUARD_TAU=1e-30 (set by the caller): the run-time guard
                                                  switches to the accurate form in mid-run; plan_products must build the stored transpose
   python3 tools/single_copy_replan.py nnls_hals  smk_solver_nnls_hals (flatclust's NnlsHals) on a solver created with a 16-bit form
Prints one JSON line with the errors against the oracle."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, smallk_amd as gpu

mode = sys.argv[1]
m, n, k, iters = 1300, 900, 24, 8          # 0 < m mod 256 <= 128: the padded-row case as well
A = oracle.fill_uniform(m, n, 51, quant=0)
W0 = oracle.fill_uniform(m, k, 52)
H0 = oracle.fill_uniform(k, n, 53) * (2.0 / k)
gpu.initialize(0)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
D = gpu.DenseMatrix.from_host(A, storage="f32", single_copy=True)
out = {"mode": mode, "single_at_start": bool(D.single_copy)}
if mode == "guard":
    s = gpu.NmfSolver(D, gpu.make_options(m, n, k, "BPP", min_iter=iters, max_iter=iters, normalize=False))
    out["form_start"] = s.product_form()[0]
    s.set_factors(W0, H0)
    rc, it, _ = s.run()
    W, H = s.factors(normalize=False)
    form, checks, fired, last = s.product_form()
    ref = oracle.nmf(A, W0, H0, "BPP", min_iter=iters, max_iter=iters, normalize=False)
    out.update(rc=rc, form_end=form, guard_checks=checks, guard_fired=fired, relW=rel(W, ref.W), relH=rel(H, ref.H),
               single_at_end=bool(D.single_copy))
else:
    from oracle import flatclust as of
    from smallk_amd import _lib as L
    import ctypes as C
    s = gpu.NmfSolver(D, gpu.make_options(m, n, k, "HALS", normalize=False))
    out["form_start"] = s.product_form()[0]
    s.set_factors(W0, H0)
    itc = C.c_int(0)
    rc = L.lib().smk_solver_nnls_hals(s._h, C.c_double(5e-2), 0, 200, C.byref(itc))
    W, H = s.factors(normalize=False)
    ok, Wr, Hr, itr = of.nnls_hals(A, W0, H0, 5e-2, 200)
    out.update(rc=rc, iterations=itc.value, ref_iterations=int(itr), form_end=s.product_form()[0], relH=rel(H, Hr), relW=rel(W, Wr),
               single_at_end=bool(D.single_copy))
print(json.dumps(out))
