"""Multi-GPU glue: one process per GPU, A and H column-sharded, W replicated.

The data path has exactly three exchange steps per iteration, all sum-all-reduces:
HH' (k x k, fp64), H*At = (A H')' (k x m, fp32) and -- when the stopping rule is evaluated --
one scalar.  The C library calls back into ``TorchAllReduce`` which runs
``torch.distributed.all_reduce`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests) on a view of the registered workspace tensor.
"""
from __future__ import annotations

import numpy as np


def shard_columns(n: int, world: int, rank: int):
    """Contiguous, balanced column ranges; the first n % world ranks get one extra column."""
    base, extra = divmod(n, world)
    ncols = base + (1 if rank < extra else 0)
    col0 = rank * base + min(rank, extra)
    return col0, ncols


def chunk_geometry(m: int, world: int, chunks: int | None = None):
    """The block-cyclic row layout of the native exchange (solver.cpp: smk_solver_attach_comm).

    The rows of A (= rows of W) are cut into ``nchunk`` chunks of ``world * blk`` rows; block r of a chunk belongs to
    rank r.  A chunk is one contiguous range of the H*At pass, the send buffer of one reduce-scatter and the receive
    buffer of one all-gather.  Returns (blk, nchunk, rows_cap); blk is a multiple of 256."""
    mpad = -(-m // 256) * 256
    c = max(1, min(mpad // (world * 4096), 4)) if chunks is None else max(1, min(int(chunks), 8))
    blk = -(-(-(-mpad // (world * c))) // 256) * 256
    nchunk = -(-mpad // (world * blk))
    return blk, nchunk, nchunk * world * blk


def own_blocks(m: int, world: int, rank: int, blk: int, nchunk: int):
    """Row ranges [a, b) of W that ``rank`` solves (valid rows only)."""
    out = []
    for j in range(nchunk):
        a = (j * world + rank) * blk
        b = min(a + blk, m)
        if b > a:
            out.append((a, b))
    return out


class TorchAllReduce:
    """Owns the comm workspace (a torch uint8 tensor on the GPU) and all-reduces views of it."""

    def __init__(self, nbytes: int, device, group=None):
        import torch
        self.torch = torch
        self.group = group
        self.ws = torch.zeros(int(nbytes) + 256, dtype=torch.uint8, device=device)
        self.base = self.ws.data_ptr()
        pad = (-self.base) % 256
        self.ptr = self.base + pad
        self.nbytes = int(nbytes)
        self._off = pad

    def __call__(self, ptr: int, count: int, dtype: int) -> int:
        import torch.distributed as dist
        torch = self.torch
        esz = 4 if dtype == 0 else 8
        off = ptr - self.base
        assert 0 <= off and off + count * esz <= self.ws.numel(), "pointer outside the comm workspace"
        view = self.ws[off:off + count * esz].view(torch.float32 if dtype == 0 else torch.float64)
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        return 0


def attach(solver, rank: int, world: int, device, group=None):
    """Register the torch-backed all-reduce with an NmfSolver.  Call before set_factors()."""
    ar = TorchAllReduce(solver.comm_workspace_bytes(), device, group)
    solver.set_comm(rank, world, ar, ar.ptr, ar.nbytes)
    solver._allreduce = ar          # keep alive
    return ar


# ---- reference semantics of the sharded algorithm on the CPU (used by the gloo tests) ----------
def sharded_hals_reference(A_local, W, H_local, iters, allreduce):
    """Column-sharded HALS written with numpy + an all-reduce callable; must equal the unsharded
    algorithm (nmf_solver_hals.hpp:166-199) up to fp64 summation order."""
    k = W.shape[1]
    W = W.copy()
    H = H_local.copy()
    HHt = allreduce(H @ H.T)
    AHt = allreduce(A_local @ H.T)
    for _ in range(iters):
        for c in range(k):
            w = W[:, c] + (AHt[:, c] - W @ HHt[:, c]) / HHt[c, c]
            w = np.where(np.isnan(w) | (w < 0), 0.0, w)
            if not w.any():
                w[:] = np.finfo(np.float64).eps
            W[:, c] = w / np.linalg.norm(w)
        WtW = W.T @ W
        WtA = W.T @ A_local
        for r in range(k):
            h = H[r, :] + (WtA[r, :] - WtW[r, :] @ H) / WtW[r, r]
            H[r, :] = np.where(np.isnan(h) | (h < 0), 0.0, h)
        HHt = allreduce(H @ H.T)
        AHt = allreduce(A_local @ H.T)
    return W, H
