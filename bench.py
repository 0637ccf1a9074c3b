#!/usr/bin/env python3
"""bench.py -- NMF iterations/sec of the MI355X dense NMF path on synthetic data.

  python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c4|c1]

A "step" is one NMF iteration (one pass of the hot path: both streaming products over A plus the
factor updates).  Default workload = BASELINE.json configs[2] ("C3", the MFMA-roofline run):
dense 65536 x 16384, k = 32, HALS, A held as bf16.  With N > 1 (launched by torch.distributed.run)
the SAME matrix is column-sharded over the ranks ("strong" scaling, as north_star asks: the named
(m,n,k) at 1/2/4/8 GPUs); exchange = RCCL all-reduce of HH' and (AH')' only.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (bigprod_kernel, both passes): algorithmic bytes per launch
                  (m * n_local * sizeof(A element)) / its average launch duration, measured live
                  with HIP events on the solver's stream, against the 8 TB/s HBM peak.
  cpu_baseline -- the CPU oracle (a port, not the reference binary) on a bounded sample of the
                  same workload, timed on this host's cores; N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (m, n, k, algorithm, storage, description)
    "c1": (512, 256, 8, "MU", "f32", "C1 dense 512x256 k=8 MU fp32 (plumbing)"),
    "c2": (8192, 4096, 16, "BPP", "f32", "C2 dense 8192x4096 k=16 BPP fp32"),
    "c3": (65536, 16384, 32, "HALS", "bf16", "C3 dense 65536x16384 k=32 HALS bf16 (MFMA roofline run)"),
    "c4": (262144, 65536, 64, "BPP", "f32", "C4 dense 262144x65536 k=64 BPP fp32"),
    # experiments (not BASELINE configs)
    "c4t": (65536, 262144, 64, "BPP", "f32", "EXPERIMENT transposed C4 shape 65536x262144 k=64 BPP fp32"),
    "c4s": (262144, 8192, 64, "BPP", "f32", "EXPERIMENT one 1/8 column shard of C4: 262144x8192 k=64 BPP fp32"),
    "c4b": (262144, 65536, 64, "BPP", "bf16", "EXPERIMENT C4 with A held as bf16"),
    "b32": (32768, 8192, 32, "BPP", "f32", "EXPERIMENT 32768x8192 k=32 BPP fp32"),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TF = {"bf16": 2500.0, "f32": 157.3}


def cpu_baseline(m, n, k, alg, quant, budget_s=20.0):
    """Time the oracle on a bounded sample (same k, algorithm, dtype rounding; fewer rows/cols)."""
    import numpy as np
    import oracle
    ms, ns = min(m, 8192), min(n, 4096)
    A = oracle.fill_uniform(ms, ns, 42, quant=quant)
    W0 = oracle.fill_uniform(ms, k, 43)
    H0 = oracle.fill_uniform(k, ns, 44) * (2.0 / k)          # same start as the GPU leg: E[W0 H0] = E[A]
    oracle.nmf(A, W0, H0, alg, min_iter=1, max_iter=1)        # warm up threads/pages
    iters, t = 2, 0.0
    while True:
        t0 = time.perf_counter()
        r = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters)
        t = time.perf_counter() - t0
        if t > budget_s / 2 or iters >= 128:
            break
        iters *= 2
    it_s_sample = r.iteration_count / t
    scale = (ms * ns) / float(m * n)
    return {
        "value": it_s_sample * scale, "unit": "iterations/s", "cores": oracle.num_threads(), "kind": "port",
        "sample": f"oracle (C/OpenMP fp64 restatement) on a {ms}x{ns} k={k} {alg} sub-problem, {r.iteration_count} "
                  f"iterations in {t:.2f} s = {it_s_sample:.3f} it/s; scaled by (sample m*n)/(full m*n) = {scale:.4g}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import smallk_amd
    from smallk_amd import dist as sdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # Test hooks for a 1-GPU box (the multi-rank path otherwise only runs on the driver's 8-GPU node):
    # SMK_BENCH_SHARE_GPU=1 puts every rank on device 0, SMK_BENCH_BACKEND=gloo replaces RCCL (which refuses
    # two ranks on one device) by gloo on device tensors.  Numbers from such a run mean nothing.
    share = os.environ.get("SMK_BENCH_SHARE_GPU", "0") == "1"
    backend = os.environ.get("SMK_BENCH_BACKEND", "nccl")
    device_index = 0 if share else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    smallk_amd.initialize(device_index)
    if world > 1:
        # the solver launches on torch's current stream so that RCCL all-reduces order against it
        smallk_amd.set_stream(torch.cuda.current_stream().cuda_stream)

    m, n, k, alg, storage, desc = WORKLOADS[args.workload]
    col0, ncols = sdist.shard_columns(n, world, rank)
    total_iters = args.warmup + args.steps

    A = smallk_amd.DenseMatrix(m, n, col0=col0, ncols=ncols, storage=storage)
    A.fill_uniform(42)
    W0 = smallk_amd.uniform_host(m, k, 43)
    # E[A] = 1/2: scale H0 so that W0 H0 has the same mean.  (An unscaled uniform start makes the first HALS
    # W update clamp every entry to zero and the run would iterate on the all-eps guard columns.)
    H0 = smallk_amd.uniform_host(k, ncols, 44, c0=col0, gheight=k) * (2.0 / k)
    opts = smallk_amd.make_options(m, n, k, alg, min_iter=total_iters, max_iter=total_iters)
    solver = smallk_amd.NmfSolver(A, opts)
    if world > 1:
        sdist.attach(solver, rank, world, dev)
    solver.set_factors(W0, H0)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    solver.iterate(args.warmup)                  # includes solver.Init
    rc = solver.sync()
    assert rc == 0, f"solver failed during warm-up: {rc}"
    torch.cuda.synchronize()
    barrier()
    solver.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.iterate(args.steps)
    rc = solver.sync()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    assert rc == 0, f"solver failed: {rc}"

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    dump = os.environ.get("SMK_BENCH_DUMP_W")        # test hook: W (replicated) after the timed steps, rank 0
    if dump:
        Wd, _ = solver.factors(normalize=False)
        if rank == 0:
            np.save(dump, Wd)

    ms0, c0 = solver.kernel_time(0)
    ms1, c1 = solver.kernel_time(1)
    bytes_per_launch, flops_per_launch = solver.kernel_work(0)
    avg_ms = (ms0 + ms1) / max(c0 + c1, 1)
    achieved_gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    mfma_tf = flops_per_launch / (avg_ms * 1e-3) / 1e12

    if rank == 0:
        out = {
            "metric": "NMF iterations/sec",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": storage if storage == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": desc, "m": m, "n": n, "k": k, "algorithm": alg, "A_storage": storage,
                       "state": "W,H,Gram fp64; big products fp32-accumulate MFMA",
                       "parallelism": f"column-shard x{world}" if world > 1 else "single GPU"},
            "mfma_tflops_big_products": mfma_tf,
            "mfma_frac_of_peak": mfma_tf / MFMA_PEAK_TF[storage],
            "whole_iteration_tflops": 4.0 * m * n * k / (elapsed / args.steps) / 1e12,
            "roofline": {
                "bound": "hbm", "kernel": "smk::bigprod_kernel (W'A and H*At passes)",
                "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None,
                "avg_launch_ms": avg_ms, "launches": c0 + c1,
                "pass_WtA_ms": ms0 / max(c0, 1), "pass_HAt_ms": ms1 / max(c1, 1),
                "algorithmic_bytes_per_launch": bytes_per_launch,
            },
        }
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                key = f"{args.workload}_n{world}"
                if key in pj:
                    out["roofline"]["traffic"] = pj[key]["bytes_per_launch"]
                    out["roofline"]["traffic_source"] = pj[key].get("source", "profiles/")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, n, k, alg, 1 if storage == "bf16" else 0)
        print(json.dumps(out), flush=True)
    barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
