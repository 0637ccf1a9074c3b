"""Throughput while the passive sets move: time of EACH iteration of a BPP run from a cold start on i.i.d.
uniform data and on data with sparse planted factors (smk_matrix_fill_planted), device-generated, any size.
   python tools/active_pivoting.py m n k iters [uniform|planted|both] [emulate_world]
Per data set: per-iteration ms (iterate(1) + sync, host clock; at C4 the sync costs < 0.1 %), it/s over iterations 1-20 and in
steady state (median of the last third).  Under rocprofv3 --kernel-trace, tools/kernel_timeline.py lists the NNLS launches in order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smallk_amd

m, n, k, iters = (int(x) for x in sys.argv[1:5])
which = sys.argv[5] if len(sys.argv) > 5 else "both"
emu = int(sys.argv[6]) if len(sys.argv) > 6 else 0
alg = os.environ.get("SMK_TOOL_ALG", "BPP")
smallk_amd.initialize(0)
comm = None
col0, ncols = 0, n
if emu > 1:
    from smallk_amd import dist as sdist
    os.environ["SMK_COMM_FORCE"] = "1"
    os.environ["SMK_COMM_EMULATE_WORLD"] = str(emu)
    col0, ncols = sdist.shard_columns(n, emu, 0)
    comm = smallk_amd.Comm.init_all(1)[0]
for data in (("uniform", "planted") if which == "both" else (which,)):
    A = smallk_amd.DenseMatrix(m, n, col0=col0, ncols=ncols)
    A.fill_uniform(42) if data == "uniform" else A.fill_planted(42, k, 0.7, 0.05)
    s = smallk_amd.NmfSolver(A, smallk_amd.make_options(m, n, k, alg, min_iter=iters, max_iter=iters))
    if comm is not None:
        s.attach_comm(comm)
    W0 = smallk_amd.uniform_host(m, k, 43)
    H0 = smallk_amd.uniform_host(k, ncols, 44, c0=col0, gheight=k)
    s.set_factors(W0, H0)
    s.iterate(0); s.sync()
    ts = []
    for i in range(iters):
        t0 = time.perf_counter(); s.iterate(1); rc = s.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0, rc
    tail = sorted(ts[2 * len(ts) // 3:])
    first20 = ts[:20]
    print(f"{m}x{n} (local columns {ncols}) k={k} {alg} {data}" + (f" emulated rank 0 of {emu}" if emu > 1 else "") +
          f": iterations 1-20: {1e3 * len(first20) / sum(first20):.1f} it/s ({sum(first20) / len(first20):.2f} ms avg, first {ts[0]:.2f} ms); "
          f"steady state (median of the last third): {1e3 / tail[len(tail) // 2]:.1f} it/s ({tail[len(tail) // 2]:.2f} ms)")
    print("   ms per iteration: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    s.close()
    A.close()
