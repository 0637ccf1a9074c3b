"""Per-GPU compute time of the column-sharded C3 iteration WITHOUT communication: rank 0's shard of
an N-way split on one GPU, the all-reduce callback a no-op (numerically meaningless, timing only).
Gives the compute part of the strong-scaling curve; add the RCCL all-reduce time of 8 MB + 8 KB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, smallk_amd
from smallk_amd import dist as sdist
smallk_amd.initialize(0)
smallk_amd.set_stream(torch.cuda.current_stream().cuda_stream)
m, n, k = 65536, 16384, 32
for world in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    col0, ncols = sdist.shard_columns(n, world, 0)
    A = smallk_amd.DenseMatrix(m, n, col0=col0, ncols=ncols, storage="bf16")
    A.fill_uniform(42)
    s = smallk_amd.NmfSolver(A, smallk_amd.make_options(m, n, k, "HALS", min_iter=100, max_iter=100))
    if world > 1:
        ws = torch.zeros(s.comm_workspace_bytes() + 256, dtype=torch.uint8, device="cuda")
        s.set_comm(0, world, lambda p, c, d: 0, ws.data_ptr() + (-ws.data_ptr()) % 256, s.comm_workspace_bytes())
    s.set_factors(smallk_amd.uniform_host(m, k, 43), smallk_amd.uniform_host(k, ncols, 44) * (2.0 / k))
    s.iterate(3); s.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); s.iterate(30); s.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"world {world}: shard {ncols} cols, {dt*1e3:.3f} ms/iter compute only", flush=True)
    s.close(); A.close()
