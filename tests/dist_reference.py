"""TEST INFRASTRUCTURE: the sharded BPP schedule of solver.cpp (native communicator) restated with numpy + a
collectives object, NNLS through the CPU oracle.  Used by the world-size-2 gloo test; mirrors, step for step:

  H_g   <- NNLS(W'W, W'A_g)                                  local columns
  HH'   <- all-reduce(H_g H_g')                               k x k
  for each row chunk j (world * blk rows, block r of a chunk belongs to rank r):
      R_j  <- (A_g H_g')[rows of chunk j]                     partial product
      own  <- reduce-scatter(R_j)                             every rank receives the sum of ITS block only
      W[own block] <- NNLS(HH', own')                         block-cyclic row ownership
  W'W   <- all-reduce(sum over own blocks of W_b' W_b)        k x k
  W     <- all-gather per chunk of the own blocks             (the product ships the packed operand; same rows)
  W'A_g <- accumulated chunk by chunk down the rows
"""
import numpy as np

import oracle
from smallk_amd.dist import chunk_geometry, own_blocks


def sharded_bpp_reference(A_local, W0, H_local, iters, coll, rank, world, chunks=None):
    m, k = W0.shape
    blk, nchunk, rows_cap = chunk_geometry(m, world, chunks)
    mine = own_blocks(m, world, rank, blk, nchunk)
    W = np.zeros((rows_cap, k))
    W[:m] = W0
    H = H_local.copy(order="F")

    def gram_w():
        g = np.zeros((k, k))
        for a, b in mine:
            g += W[a:b].T @ W[a:b]
        return coll.allreduce(g)

    def gather_w():
        for j in range(nchunk):
            r0 = j * world * blk
            W[r0:r0 + world * blk] = coll.allgather(W[r0 + rank * blk:r0 + (rank + 1) * blk])

    def wta():
        acc = np.zeros((k, A_local.shape[1]))
        for j in range(nchunk):
            r0, r1 = j * world * blk, min((j + 1) * world * blk, m)
            if r1 > r0:
                acc += W[r0:r1].T @ A_local[r0:r1]
        return acc

    WtW = gram_w()
    WtA = wta()
    for _ in range(iters):
        ok, X, _, _ = oracle.nnls_blockpivot(WtW, WtA, H)
        assert ok
        H = X
        HHt = coll.allreduce(H @ H.T)
        for j in range(nchunk):
            r0 = j * world * blk
            part = np.zeros((world * blk, k))
            r1 = min(r0 + world * blk, m)
            if r1 > r0:
                part[:r1 - r0] = A_local[r0:r1] @ H.T
            own = coll.reduce_scatter(part, blk)                 # blk x k: the sum over ranks of block `rank`
            a = r0 + rank * blk
            b = min(a + blk, m)
            if b > a:
                ok, X, _, _ = oracle.nnls_blockpivot(HHt, np.asfortranarray(own[:b - a].T), np.asfortranarray(W[a:b].T))
                assert ok
                W[a:b] = X.T
        WtW = gram_w()
        gather_w()
        WtA = wta()
    return W[:m].copy(), H


def sharded_mu_reference(A_local, W0, H_local, iters, coll, rank, world, chunks=None):
    """The same exchange with the multiplicative updates (nmf_solver_mu.hpp:121-164, solver.cpp MU schedule with a row-sharded W):
    H_g <- H_g .* (W'A_g) ./ (W'W H_g + 1e-13) on the local columns; every rank updates the rows of W it owns,
    W_b <- W_b .* own ./ (W_b HH' + 1e-13), from its reduce-scattered block of (A H')'."""
    m, k = W0.shape
    blk, nchunk, rows_cap = chunk_geometry(m, world, chunks)
    mine = own_blocks(m, world, rank, blk, nchunk)
    W = np.zeros((rows_cap, k))
    W[:m] = W0
    H = H_local.copy()

    def gram_w():
        g = np.zeros((k, k))
        for a, b in mine:
            g += W[a:b].T @ W[a:b]
        return coll.allreduce(g)

    def gather_w():
        for j in range(nchunk):
            r0 = j * world * blk
            W[r0:r0 + world * blk] = coll.allgather(W[r0 + rank * blk:r0 + (rank + 1) * blk])

    def wta():
        acc = np.zeros((k, A_local.shape[1]))
        for j in range(nchunk):
            r0, r1 = j * world * blk, min((j + 1) * world * blk, m)
            if r1 > r0:
                acc += W[r0:r1].T @ A_local[r0:r1]
        return acc

    WtW = gram_w()
    WtA = wta()
    for _ in range(iters):
        H = H * (WtA / (WtW @ H + 1.0e-13))
        HHt = coll.allreduce(H @ H.T)
        for j in range(nchunk):
            r0 = j * world * blk
            part = np.zeros((world * blk, k))
            r1 = min(r0 + world * blk, m)
            if r1 > r0:
                part[:r1 - r0] = A_local[r0:r1] @ H.T
            own = coll.reduce_scatter(part, blk)
            a = r0 + rank * blk
            b = min(a + blk, m)
            if b > a:
                W[a:b] = W[a:b] * (own[:b - a] / (W[a:b] @ HHt + 1.0e-13))
        WtW = gram_w()
        gather_w()
        WtA = wta()
    return W[:m].copy(), H

