#!/usr/bin/env python3
"""bench.py -- NMF iterations/sec of the MI355X dense NMF path on synthetic data.

  python bench.py --gpus N --steps K --warmup W [--workload c4|c3|c2|c1]

A "step" is one NMF iteration (one pass of the hot path: both streaming products over A plus the
factor updates).  Default workload = BASELINE.json configs[3] ("C4"): dense 262144 x 65536, k = 64, BPP, fp32 --
the configuration north_star's 1/2/4/8-GPU throughput and >= 6x target are quoted on, and the largest one that
fits a single MI355X (137 GB of A and A').  `--workload c3` is configs[2] (65536 x 16384, k = 32, HALS, bf16 A,
the MFMA-roofline run; its line is kept in profiles/), c2 is configs[1].  With N > 1 (launched by
torch.distributed.run) the SAME matrix is column-sharded over the ranks ("strong" scaling: the named (m,n,k) at
1/2/4/8 GPUs); exchange = RCCL all-reduce of HH' and (AH')' (+ an all-gather of W for BPP), issued from C by
libsmallk_amd.so on the solver's streams -- no Python inside the iteration loop.  torch.distributed (gloo, CPU)
is used only to broadcast the RCCL unique id and for the timing barrier.

The K timed steps are one window; the window is repeated (5 times, and until >= 0.5 s have been timed) and
the MEDIAN window is reported, so `value`, `ms_per_step` are per K steps as the contract asks while short
windows no longer decide the number (`windows_ms` lists them all).

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
    """The CPU oracle on a bounded sample (same k, algorithm, dtype rounding; fewer rows/cols), timed on this
    host's cores, priced so that the parts add up:

      t_iter  = t_big + t_other     per iteration of the sample, both measured INSIDE the oracle's iterations
                                    (t_big: its products with A, exported by orc_big_product_time; t_other: the rest --
                                    NNLS / element-wise updates / Gram matrices / gradients)
      t_blas  = the same products the selected algorithm runs (BPP: W'A and H*At against the stored transpose;
                MU / HALS: W'A and A*H') through torch's CPU matmul (MKL, fp64) on warm buffers: the reference links
                an optimized BLAS (sphinx/source/pages_installation.rst:355), the oracle's GEMM is a plain OpenMP loop
      value   = 1 / ( min(t_big, t_blas) * (m n)/(ms ns)  +  t_other * (m + n)/(ms + ns) )
                the products scale with the matrix, the rest with the number of factor rows and columns."""
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
    t_big_total, big_calls = oracle.big_product_time()
    t_iter = t / r.iteration_count
    t_big = 2.0 * t_big_total / max(big_calls, 1)             # two products per iteration (+ one in Init)
    t_other = max(t_iter - t_big, 0.0)
    kind, blas, t_blas = "port", "none (oracle's OpenMP loops)", t_big
    try:
        import torch
        torch.set_num_threads(oracle.num_threads())
        At = torch.from_numpy(A.T).T                          # column-major views, no copies
        Wt_, Ht = torch.from_numpy(W0.T).T, torch.from_numpy(H0.T).T
        if alg == "BPP":
            Att = torch.from_numpy(np.asfortranarray(A.T).T).T          # the stored transpose (nmf_solver_bpp.hpp:319)
            run = lambda: ((Wt_.T @ At), (Ht @ Att))
        else:
            run = lambda: ((Wt_.T @ At), (At @ Ht.T))
        time.sleep(0.2)                                       # let the oracle's OpenMP team go to sleep
        run()
        best = float("inf")
        for _ in range(5):
            t0 = time.perf_counter()
            run()
            best = min(best, time.perf_counter() - t0)
        t_blas = best
        blas = "torch CPU matmul fp64 (" + str(torch.__config__.show().split("BLAS_INFO=")[-1].split(",")[0]).strip() + ")"
        kind = "port+blas"
    except Exception:
        pass
    t_prod = min(t_big, t_blas)
    s_prod = (m * n) / float(ms * ns)
    s_other = (m + n) / float(ms + ns)
    t_full = t_prod * s_prod + t_other * s_other
    return {
        "value": 1.0 / t_full, "unit": "iterations/s", "cores": oracle.num_threads(), "kind": kind, "blas": blas,
        "sample_it_s": 1.0 / (t_prod + t_other), "sample_it_s_plain_oracle": 1.0 / t_iter,
        "sample_ms": {"iteration": t_iter * 1e3, "big_products_inside": t_big * 1e3, "other": t_other * 1e3,
                      "big_products_blas": t_blas * 1e3, "priced_products": t_prod * 1e3},
        "scale": {"products_mn": s_prod, "other_m_plus_n": s_other},
        "sample": f"oracle (C/OpenMP fp64 restatement) on a {ms}x{ns} k={k} {alg} sub-problem: {r.iteration_count} iterations in "
                  f"{t:.2f} s = {t_iter * 1e3:.1f} ms each = {t_big * 1e3:.1f} ms in its two products with A + {t_other * 1e3:.1f} ms "
                  f"elsewhere (NNLS / updates / Gram); the same two products through {blas}: {t_blas * 1e3:.1f} ms; priced "
                  f"{t_prod * 1e3:.1f} ms x {s_prod:.4g} (m n ratio) + {t_other * 1e3:.1f} ms x {s_other:.4g} ((m + n) ratio) "
                  f"= {t_full:.3f} s per full-size iteration",
    }


def kernel_source_sha16():
    """first 16 hex digits of sha256(smallk_amd/csrc/bigprod.hip): ties a PMC traffic figure to the kernel it was taken on"""
    import hashlib
    with open(os.path.join(ROOT, "smallk_amd", "csrc", "bigprod.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="MEASUREMENT HOOK, one GPU: run as rank 0 of a world of this many ranks -- its column shard, the chunk "
                         "geometry and the row blocks of that world, every collective issued through RCCL with ONE rank "
                         "(device-local).  Times the per-rank work of an N-GPU run without the xGMI transfers; the factors "
                         "it produces are meaningless (the other ranks' blocks never arrive)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import smallk_amd
    from smallk_amd import dist as sdist

    if args.emulate_world > 1:
        os.environ["SMK_COMM_FORCE"] = "1"
        os.environ["SMK_COMM_EMULATE_WORLD"] = str(args.emulate_world)
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
    native = backend == "nccl"          # default: RCCL from C (comm.cpp); "gloo": the round-1 callback hook (tests)
    rccl_defaults_set = []              # environment defaults this script added (taken back for plan B)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")                 # CPU side channel only: unique id, barrier, max of the clocks
        # RCCL's own description of what it built (rings / trees / transport per channel) goes to a file per rank; rank 0
        # prints an excerpt to stderr after the run so that a scaling run explains itself.  The caller's settings win.
        if "NCCL_DEBUG" not in os.environ:
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH" + (",TUNING" if os.environ.get("SMK_BENCH_RCCL_TUNING") else ""))
            os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/smk_rccl_%h_%p.log")
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            # one node: RCCL's bootstrap sockets stay on the loopback interface (the container's hostname may not
            # resolve) and no InfiniBand probing; the data path is xGMI either way.  The caller's settings win.
            for key, val in (("NCCL_SOCKET_IFNAME", "lo"), ("NCCL_IB_DISABLE", "1")):
                if key not in os.environ:
                    os.environ[key] = val
                    rccl_defaults_set.append(key)
    smallk_amd.initialize(device_index)
    comm, fallback_group, collectives = None, None, "none"
    # RCCL prints a version banner on stdout when a communicator is created: keep stdout for the one JSON line
    saved_stdout = None
    if world > 1 or args.emulate_world > 1:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    if world > 1 and native:
        # RCCL communicator created by libsmallk_amd.so itself; every collective of the iteration is issued from C
        ok = 1
        try:
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.tensor(list(smallk_amd.Comm.unique_id()), dtype=torch.uint8)
            torch.distributed.broadcast(uid, 0)
            comm = smallk_amd.Comm.init_rank(bytes(uid.tolist()), rank, world)
            comm.selftest()                 # known sums through RCCL before the run is trusted to it
        except Exception as e:              # pragma: no cover  (multi-GPU nodes only)
            print(f"[bench rank {rank}] native RCCL communicator failed: {e}", file=sys.stderr, flush=True)
            ok = 0
        agreed = torch.tensor([ok], dtype=torch.int32)
        torch.distributed.all_reduce(agreed, op=torch.distributed.ReduceOp.MIN)
        if int(agreed.item()) == 1:
            collectives = "RCCL from C (comm.cpp)"
        else:                               # plan B, all ranks together: torch's own RCCL through the callback hook
            if comm is not None:
                comm.close()
                comm = None
            native = False
            for key in rccl_defaults_set:   # plan B runs with RCCL's own defaults
                os.environ.pop(key, None)
            fallback_group = torch.distributed.new_group(backend="nccl")
            collectives = "torch.distributed nccl through the callback hook (native communicator failed)"
    elif world > 1:
        collectives = "callback hook (" + backend + ")"
    if world > 1 and not native:
        # callback hook: the solver launches on torch's current stream so that the all-reduces order against it
        smallk_amd.set_stream(torch.cuda.current_stream().cuda_stream)
    m, n, k, alg, storage, desc = WORKLOADS[args.workload]
    col0, ncols = sdist.shard_columns(n, world, rank)
    if args.emulate_world > 1:
        assert world == 1, "--emulate-world is a one-GPU hook"
        col0, ncols = sdist.shard_columns(n, args.emulate_world, 0)
        comm = smallk_amd.Comm.init_all(1)[0]
        collectives = f"EMULATED rank 0 of {args.emulate_world}: RCCL calls with one rank (device-local)"
    def restore_stdout():
        """fd 1 was pointed at stderr while RCCL could print (its banner at communicator creation, INFO lines at the first
        collectives if NCCL_DEBUG_FILE is not honoured): stdout carries ONLY the JSON line, so it comes back right before it"""
        nonlocal saved_stdout
        if saved_stdout is None:
            return
        sys.stdout.flush()
        try:                                # what sits in the C library's stdio buffer goes out while fd 1 is still stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
        saved_stdout = None

    total_iters = args.warmup + args.steps
    A = smallk_amd.DenseMatrix(m, n, col0=col0, ncols=ncols, storage=storage)
    A.fill_uniform(42)
    W0 = smallk_amd.uniform_host(m, k, 43)
    # E[A] = 1/2: scale H0 so that W0 H0 has the same mean.  (An unscaled uniform start makes the first HALS
    # W update clamp every entry to zero and the run would iterate on the all-eps guard columns.)
    H0 = smallk_amd.uniform_host(k, ncols, 44, c0=col0, gheight=k) * (2.0 / k)
    opts = smallk_amd.make_options(m, n, k, alg, min_iter=total_iters, max_iter=total_iters)
    solver = smallk_amd.NmfSolver(A, opts)
    if comm is not None:
        solver.attach_comm(comm)
    elif world > 1:
        sdist.attach(solver, rank, world, dev, group=fallback_group)
    solver.set_factors(W0, H0)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    solver.iterate(args.warmup)                  # includes solver.Init
    rc = solver.sync()
    assert rc == 0, f"solver failed during warm-up: {rc}"
    torch.cuda.synchronize()
    solver.enable_timing(True)

    def window():
        """EXACTLY args.steps iterations between barrier + synchronize pairs; max over ranks."""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.iterate(args.steps)
        rcw = solver.sync()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        assert rcw == 0, f"solver failed: {rcw}"
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    test_hook = bool(os.environ.get("SMK_BENCH_DUMP_W"))          # the 2-process test compares W after exactly K steps
    windows = [window()]
    while not test_hook and (len(windows) < 5 or sum(windows) < 0.5) and len(windows) < 200:
        windows.append(window())
    elapsed = sorted(windows)[len(windows) // 2]

    dump = os.environ.get("SMK_BENCH_DUMP_W")        # test hook: W (replicated) after the timed steps, rank 0
    if dump:
        Wd, _ = solver.factors(normalize=False)
        if rank == 0:
            np.save(dump, Wd)

    ms0, c0 = solver.kernel_time(0)
    ms1, c1 = solver.kernel_time(1)
    sharded = world > 1 or args.emulate_world > 1
    msc, cc = solver.kernel_time(2) if sharded else (0.0, 0)    # spans of the collectives on the second stream, this rank
    bytes_per_launch, flops_per_launch = solver.kernel_work(0)
    avg_ms = (ms0 + ms1) / max(c0 + c1, 1)
    achieved_gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    mfma_tf = flops_per_launch / (avg_ms * 1e-3) / 1e12
    # per rank and per step: both streaming passes, the collectives' own spans, and what the step spends outside the
    # passes.  If the collectives take longer than that remainder, the difference was hidden behind the products:
    # overlap_lower_bound = max(0, 1 - outside / collectives).
    timed_steps = max(len(windows) * args.steps, 1)
    ngroups = max((k + 63) // 64, 1)
    per_rank = {"rank": rank, "ms_per_step": sum(windows) / timed_steps * 1e3,
                "products_ms_per_step": (ms0 + ms1) / timed_steps, "collectives_ms_per_step": msc / timed_steps,
                "collective_calls_per_step": cc / timed_steps, "passes_per_step": (c0 + c1) / ngroups / timed_steps}
    per_rank["outside_products_ms_per_step"] = per_rank["ms_per_step"] - per_rank["products_ms_per_step"]
    per_rank["overlap_lower_bound"] = (max(0.0, 1.0 - per_rank["outside_products_ms_per_step"] / per_rank["collectives_ms_per_step"])
                                       if per_rank["collectives_ms_per_step"] > 0 else None)
    ranks_report = [per_rank]
    if world > 1:
        ranks_report = [None] * world
        torch.distributed.all_gather_object(ranks_report, per_rank)

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
                       "state": "W,H,Gram fp64; big products: MFMA with fp32 accumulation folded into fp64",
                       "parallelism": (f"column-shard x{world}" if world > 1 else
                                       f"EMULATED rank 0 of {args.emulate_world} on one GPU (not a benchmark line)" if args.emulate_world > 1
                                       else "single GPU"),
                       "collectives": collectives},
            # useful flops (2 k per matrix entry) of the streaming products.  fp32 storage computes them as three fp16
            # MFMAs per product, so its ratio is against the NATIVE fp32 matrix peak that this replaces
            # and may exceed 1; the bound that matters for this path is roofline.frac (HBM).
            "useful_tflops_big_products": mfma_tf,
            "useful_tflops_vs_native_mfma_peak": mfma_tf / MFMA_PEAK_TF[storage],
            "whole_iteration_tflops": 4.0 * m * n * k / (elapsed / args.steps) / 1e12,
            "collectives_ms_per_step_rank0": per_rank["collectives_ms_per_step"] if sharded else None,
            "per_rank": ranks_report if sharded else None,
            "windows": len(windows), "windows_ms": [round(w * 1e3, 4) for w in windows],
            "timed_region_s": sum(windows),
            "roofline": {
                "bound": "hbm",
                "kernel": ("smk::bigprod_kernel" if storage == "bf16" else "smk::bigprod_f3_kernel (fp32 A as two fp16 terms)")
                          + " (W'A and H*At passes)",
                "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None,
                "avg_launch_ms": avg_ms, "launches": c0 + c1,
                "pass_WtA_ms": ms0 / max(c0, 1), "pass_HAt_ms": ms1 / max(c1, 1),
                "algorithmic_bytes_per_launch": bytes_per_launch,
            },
        }
        # HBM traffic comes from separate rocprofv3 --pmc passes (tools/profile_r03.sh -> profiles/hbm_traffic.json).  The
        # entry records the hash of the kernel source it was measured on: a figure from an older kernel is not printed.
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof) and args.emulate_world <= 1:
            try:
                pj = json.load(open(prof))
                key = f"{args.workload}_n{world}"
                if key in pj:
                    if pj[key].get("kernel_source_sha16") == kernel_source_sha16():
                        out["roofline"]["traffic"] = pj[key]["bytes_per_launch"]
                        out["roofline"]["traffic_source"] = pj[key].get("source", "profiles/")
                    else:
                        out["roofline"]["traffic_stale"] = ("profiles/hbm_traffic.json was measured on another version of "
                                                            "smallk_amd/csrc/bigprod.hip; rerun tools/profile_r03.sh")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, n, k, alg, 1 if storage == "bf16" else 0)
        restore_stdout()
        print(json.dumps(out), flush=True)
    if rank == 0 and world > 1:
        import glob
        import re
        pat = re.compile(r"Channel|Ring|Tree|Trees|XGMI|xgmi|P2P|SHM|NET|algo|Algo|proto|Connected|nChannels|comm 0x", re.I)
        for fn in sorted(glob.glob("/tmp/smk_rccl_*.log"))[:1]:
            try:
                lines = [l.rstrip() for l in open(fn, errors="replace") if pat.search(l)]
                print(f"[bench] RCCL log excerpt ({fn}, {len(lines)} matching lines, first 60):", file=sys.stderr)
                for l in lines[:60]:
                    print("[rccl] " + l[:220], file=sys.stderr)
            except Exception as e:      # pragma: no cover
                print(f"[bench] no RCCL log: {e}", file=sys.stderr)
    barrier()
    if comm is not None:
        solver.close()
        comm.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
