"""Streaming rate of the accurate rank-2 product (bigprod_f64_k2_kernel) inside dense RANK2 iterations:
   python3 tools/r2_dense_rate.py [m] [n] [storage] [iters]     (default 65536 x 16384 bf16 = the C3 matrix)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smallk_amd
m = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
storage = sys.argv[3] if len(sys.argv) > 3 else "bf16"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
smallk_amd.initialize(0)
A = smallk_amd.DenseMatrix(m, n, storage=storage)
A.fill_uniform(42)
s = smallk_amd.NmfSolver(A, smallk_amd.make_options(m, n, 2, "RANK2", min_iter=10 ** 6, max_iter=10 ** 6))
s.set_factors(smallk_amd.uniform_host(m, 2, 43), smallk_amd.uniform_host(2, n, 44))
s.iterate(3); assert s.sync() == 0
os.environ["SMK_TIMING_STRIDE"] = "1"
s.enable_timing(True)
t0 = time.perf_counter(); s.iterate(iters); assert s.sync() == 0; dt = time.perf_counter() - t0
ms0, c0 = s.kernel_time(0); ms1, c1 = s.kernel_time(1)
b, _ = s.kernel_work(0)
print(f"dense RANK2 {m} x {n} {storage}: {dt / iters * 1e3:.3f} ms per iteration; W'A pass {ms0 / max(c0, 1):.3f} ms = {b / (ms0 / max(c0, 1) * 1e-3) / 1e12:.2f} TB/s, "
      f"H*At pass {ms1 / max(c1, 1):.3f} ms = {b / (ms1 / max(c1, 1) * 1e-3) / 1e12:.2f} TB/s (product form {s.product_form()[0]})")
