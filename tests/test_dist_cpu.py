"""CPU tests of the N > 1 path (world_size 2, gloo): the column-sharded algorithm with
all-reduces of HH', AH' (and nothing else) equals the unsharded oracle; shard bookkeeping."""
import os
import socket

import numpy as np
import pytest

import oracle
from smallk_amd import dist as sdist


def test_shard_columns_partition():
    for n in (1, 7, 16, 100, 16384, 65537):
        for world in (1, 2, 3, 8):
            ranges = [sdist.shard_columns(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0
            assert sum(nc for _, nc in ranges) == n
            for (c0, nc), (c1, _) in zip(ranges, ranges[1:]):
                assert c0 + nc == c1
            assert max(nc for _, nc in ranges) - min(nc for _, nc in ranges) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m, n, k, iters, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c0, nc = sdist.shard_columns(n, world, rank)
        # every rank regenerates its own shard from the counter-based generator
        A_loc = oracle.fill_uniform(m, nc, 42, c0=c0, gheight=m)
        W0 = oracle.fill_uniform(m, k, 43)
        H_loc = oracle.fill_uniform(k, nc, 44, c0=c0, gheight=k)

        calls = []

        def allreduce(x):
            t = torch.from_numpy(np.ascontiguousarray(x))
            dist.all_reduce(t)
            calls.append(x.shape)
            return t.numpy()

        W, H = sdist.sharded_hals_reference(A_loc, W0, H_loc, iters, allreduce)
        q.put((rank, c0, nc, W, H, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_sharded_hals_equals_unsharded_oracle(world):
    import torch.multiprocessing as mp
    m, n, k, iters = 96, 50, 5, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, n, k, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    ref = oracle.nmf(A, W0, H0, "HALS", min_iter=iters, max_iter=iters, normalize=False)
    H = np.concatenate([r[4] for r in results], axis=1)
    for r in results:                                   # W is replicated and identical on every rank
        assert np.allclose(r[3], ref.W, rtol=1e-10, atol=1e-13)
    assert np.allclose(H, ref.H, rtol=1e-10, atol=1e-13)
    # the only exchanged objects are k x k and m x k
    shapes = set(results[0][5])
    assert shapes == {(k, k), (m, k)}
    assert len(results[0][5]) == 2 * (iters + 1)


def _bpp_worker(rank, world, port, m, n, k, iters, chunks, q, alg="BPP"):
    import torch
    import torch.distributed as dist
    from dist_reference import sharded_bpp_reference, sharded_mu_reference
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c0, nc = sdist.shard_columns(n, world, rank)
        A_loc = oracle.fill_uniform(m, nc, 42, c0=c0, gheight=m)
        W0 = oracle.fill_uniform(m, k, 43)
        H_loc = oracle.fill_uniform(k, nc, 44, c0=c0, gheight=k) * (2.0 / k)
        calls = []

        class Coll:
            def allreduce(self, x):
                t = torch.from_numpy(np.ascontiguousarray(x))
                dist.all_reduce(t)
                calls.append(("allreduce", x.shape))
                return t.numpy()

            def reduce_scatter(self, buf, blk):
                # gloo has no reduce-scatter: one reduce per destination block (the same data movement)
                calls.append(("reduce_scatter", buf.shape))
                own = None
                for r in range(world):
                    t = torch.from_numpy(np.ascontiguousarray(buf[r * blk:(r + 1) * blk]))
                    dist.reduce(t, dst=r)
                    if r == rank:
                        own = t.numpy().copy()
                return own

            def allgather(self, block):
                t = torch.from_numpy(np.ascontiguousarray(block))
                out = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(out, t)
                calls.append(("allgather", block.shape))
                return np.concatenate([o.numpy() for o in out], axis=0)

        W, H = (sharded_bpp_reference if alg == "BPP" else sharded_mu_reference)(A_loc, W0, H_loc, iters, Coll(), rank, world, chunks)
        q.put((rank, c0, nc, W, H, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m,n,chunks", [(700, 45, 1), (1100, 37, 2), (513, 64, 3)])
def test_sharded_bpp_exchange_equals_unsharded_oracle(m, n, chunks):
    """world size 2 over gloo: the BPP exchange of the native path -- reduce-scatter of (AH')' by row chunks, block-cyclic
    NNLS of W, W'W from the own blocks + all-reduce, all-gather of the solved blocks, W'A accumulated chunk by chunk --
    reproduces the unsharded oracle; uneven m (blocks of 256 rows: short and empty blocks) and uneven n."""
    import torch.multiprocessing as mp
    from smallk_amd.dist import chunk_geometry, own_blocks
    world, k, iters = 2, 6, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bpp_worker, args=(r, world, port, m, n, k, iters, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    ref = oracle.nmf(A, W0, H0, "BPP", min_iter=iters, max_iter=iters, normalize=False)
    H = np.concatenate([r[4] for r in results], axis=1)
    for r in results:
        assert np.linalg.norm(r[3] - ref.W) / np.linalg.norm(ref.W) < 1e-9
    assert np.linalg.norm(H - ref.H) / np.linalg.norm(ref.H) < 1e-9
    assert np.array_equal(results[0][3], results[1][3])            # the gathered W is the same on both ranks, bit for bit
    blk, nchunk, cap = chunk_geometry(m, world, chunks)
    assert blk % 256 == 0 and cap >= m and cap == nchunk * world * blk
    rows = sorted(r for rk in range(world) for a, b in own_blocks(m, world, rk, blk, nchunk) for r in range(a, b))
    assert rows == list(range(m))                                  # every row of W has exactly one owner
    kinds = [c[0] for c in results[0][5]]
    assert kinds.count("reduce_scatter") == nchunk * iters and kinds.count("allgather") == nchunk * (iters + 0)
    # exchanged objects: k x k sums, (world * blk) x k row chunks, blk x k blocks -- nothing of the size of A
    assert {c[1] for c in results[0][5]} <= {(k, k), (world * blk, k), (blk, k)}


@pytest.mark.parametrize("m,n,chunks", [(700, 45, 1), (1030, 51, 2)])
def test_sharded_mu_exchange_equals_unsharded_oracle(m, n, chunks):
    """world size 2 over gloo: MU with the W update row-sharded like block pivoting's (round 3) -- the same reduce-scatter /
    all-reduce / all-gather choreography, the multiplicative rule on the own blocks -- reproduces the unsharded oracle."""
    import torch.multiprocessing as mp
    world, k, iters = 2, 5, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bpp_worker, args=(r, world, port, m, n, k, iters, chunks, q, "MU")) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    ref = oracle.nmf(A, W0, H0, "MU", min_iter=iters, max_iter=iters, normalize=False)
    H = np.concatenate([r[4] for r in results], axis=1)
    for r in results:
        assert np.linalg.norm(r[3] - ref.W) / np.linalg.norm(ref.W) < 1e-9
    assert np.linalg.norm(H - ref.H) / np.linalg.norm(ref.H) < 1e-9
    assert np.array_equal(results[0][3], results[1][3])

