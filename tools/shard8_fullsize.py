import os, sys, threading
sys.path.insert(0, "/root/repo")
import numpy as np
import smallk_amd
from smallk_amd import DenseMatrix, NmfSolver, Comm, make_options, uniform_host, thread_context_begin, thread_context_end
from smallk_amd import dist as sdist
smallk_amd.initialize(0)
m, n, k, world = 262144, 65536, 64, 8
iters = int(sys.argv[1])
W0 = uniform_host(m, k, 312); H0 = uniform_host(k, n, 313) * (2.0 / k)
opts = dict(min_iter=iters, max_iter=iters, normalize=False)
A = DenseMatrix(m, n); A.fill_uniform(311)
s = NmfSolver(A, make_options(m, n, k, "BPP", **opts)); s.set_factors(W0, H0); s.iterate(iters); assert s.sync() == 0
W1, H1 = s.factors(normalize=False); s.close(); A.close()
comms = Comm.init_local(world); out = [None] * world
def run(rank):
    thread_context_begin(0)
    c0, nc = sdist.shard_columns(n, world, rank)
    D = DenseMatrix(m, n, col0=c0, ncols=nc); D.fill_uniform(311)
    sv = NmfSolver(D, make_options(m, n, k, "BPP", **opts)); sv.attach_comm(comms[rank]); sv.set_factors(W0, H0[:, c0:c0 + nc])
    sv.iterate(iters); rc = sv.sync(); W, H = sv.factors(normalize=False); out[rank] = (rc, W if rank == 0 else None, H); sv.close(); D.close(); thread_context_end()
ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
H8 = np.concatenate([o[2] for o in out], axis=1)
fro = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print(f"iters {iters} F64={os.environ.get('SMK_COMM_F64','0')} NSPLIT={os.environ.get('SMK_NSPLIT','-')}: relH {fro(H8, H1):.3e} relW {fro(out[0][1], W1):.3e}", flush=True)
