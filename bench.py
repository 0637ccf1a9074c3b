#!/usr/bin/env python3
"""bench.py -- NMF iterations/sec of the MI355X dense NMF path on synthetic data.

  python bench.py --gpus N --steps K --warmup W [--workload c4|c3|c2|c1] [--single-process]

A "step" is one NMF iteration (one pass of the hot path: both streaming products over A plus the
factor updates).  Default workload = BASELINE.json configs[3] ("C4"): dense 262144 x 65536, k = 64, BPP, fp32 --
the configuration north_star's 1/2/4/8-GPU throughput and >= 6x target are quoted on, and the largest one that
fits a single MI355X (137 GB of A and A').  `--workload c3` is configs[2] (65536 x 16384, k = 32, HALS, bf16 A,
the MFMA-roofline run; its line is kept in profiles/), c2 is configs[1].

N > 1: the SAME matrix is column-sharded over the ranks ("strong" scaling: the named (m,n,k) at 1/2/4/8 GPUs).
The exchange per iteration is issued from C by libsmallk_amd.so on the solver's second stream -- no Python inside
the iteration loop: an all-reduce of HH' (k x k); the sum of (AH')' pipelined in row chunks behind the H*At pass
(all-reduce for HALS, REDUCE-SCATTER for BPP / MU, whose rows of W are independent problems that each rank solves
for its own row blocks); an all-reduce of W'W; and an all-gather per chunk of the PACKED streaming operand of the
own blocks behind the W'A pass (the fp64 rows of W are gathered only when results are read).

How N ranks come to exist (`--gpus N`, N > 1):
  * under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (WORLD_SIZE set): this
    process is one rank;
  * as plain `python bench.py --gpus N ...` (WORLD_SIZE unset): this process NEVER touches a GPU.  It starts the
    ranks as a child process group (torch.distributed.run on 127.0.0.1 and a free port), relays rank 0's single
    JSON line and exits with the children's worst return code.  A watchdog follows the ranks' heartbeat: a run whose
    rendezvous, communicator set-up or first collective stalls is killed (its own process group only), the RCCL
    log excerpt is printed, and -- unless --no-fallback -- the second, independent path is tried:
  * `--single-process`: ONE process drives all N devices, one host thread per device, communicators from
    ncclCommInitAll (smk_comm_init_all) -- no torch.distributed, no rendezvous.
torch.distributed (gloo, CPU) is used only to broadcast the RCCL unique id and for the timing barrier.

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
import threading
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
    # two ranks of these have exactly the per-rank shard geometry of the 8-GPU runs of C4 / C3 (tests/test_gpu_dist.py)
    "c4s2": (262144, 16384, 64, "BPP", "f32", "EXPERIMENT two 1/8 column shards of C4: 262144x16384 k=64 BPP fp32 (--gpus 2: the 8-GPU shard geometry per rank)"),
    "c3s2": (65536, 4096, 32, "HALS", "bf16", "EXPERIMENT two 1/8 column shards of C3: 65536x4096 k=32 HALS bf16 (--gpus 2: the 8-GPU shard geometry per rank)"),
    "c4b": (262144, 65536, 64, "BPP", "bf16", "EXPERIMENT C4 with A held as bf16"),
    "b32": (32768, 8192, 32, "BPP", "f32", "EXPERIMENT 32768x8192 k=32 BPP fp32"),
    "c3t": (16384, 65536, 32, "HALS", "bf16", "EXPERIMENT C3 transposed: 16384x65536 k=32 HALS bf16"),
    "mall": (65536, 1024, 32, "HALS", "bf16", "EXPERIMENT 65536x1024 k=32 HALS bf16: 134 MB of A, both passes from the Infinity Cache with --single-copy"),
    "mall2": (65536, 2048, 32, "HALS", "bf16", "EXPERIMENT 65536x2048 k=32 HALS bf16: 268 MB of A"),
    "c4x2": (262144, 131072, 64, "BPP", "f32", "EXPERIMENT twice C4: 262144x131072 k=64 BPP fp32 -- 137 GB as a single copy (--single-copy)"),
    "c2mu": (8192, 4096, 16, "MU", "f32", "EXPERIMENT C2's matrix under MU"),
    "c3f": (65536, 16384, 32, "HALS", "f32", "EXPERIMENT C3's shape with fp32 A (HALS, bf16x3 products)"),
    "c4mu": (262144, 65536, 64, "MU", "f32", "EXPERIMENT C4's matrix under MU (replicated W update, all-reduce of (AH')')"),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TF = {"bf16": 2500.0, "f32": 157.3}

from bench_launch import (_hb, beat, lim, launch, rccl_choices, rccl_env_defaults, rccl_log_excerpt, start_rank_watchdog,
                          StdoutGuard)


def cpu_baseline(m, n, k, alg, quant, budget_s=20.0, data="uniform"):
    """cpu_baseline_at() at 16 threads (the oracle's default; the reference's sample runs use 8, pages_smallkAPI.rst:58-92) and, on a
    host with more hardware threads, at min(threads, 64) as well -- the faster of the two is reported, both are listed: `cores` is
    what the reported number actually used (round 6: the pool's boxes have 256 hardware threads)."""
    import oracle
    tried = {}
    counts = [min(os.cpu_count() or 1, 16)]
    if (os.cpu_count() or 1) >= 32 and not os.environ.get("ORACLE_THREADS"):
        counts.append(min(os.cpu_count(), 64))
    best = None
    for t in counts:
        if not os.environ.get("ORACLE_THREADS"):
            oracle.set_num_threads(t)
        r = cpu_baseline_at(m, n, k, alg, quant, budget_s / len(counts), data)
        tried[str(r["cores"])] = r["value"]
        if best is None or r["value"] > best["value"]:
            best = r
    best["thread_counts_tried_it_s"] = tried
    oracle.set_num_threads(counts[0])
    return best


def cpu_baseline_at(m, n, k, alg, quant, budget_s=20.0, data="uniform"):
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
    A = oracle.fill_uniform(ms, ns, 42, quant=quant) if data == "uniform" else oracle.fill_planted(ms, ns, 42, k, quant=quant)
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS) + ["s_reuters", "s_reuters_hals", "s_1m"],
                    help="c1..c4: BASELINE.json's dense configurations; s_*: sparse A, one GPU (bench_sparse.py)")
    ap.add_argument("--data", default="uniform", choices=["uniform", "planted"],
                    help="dense workloads: i.i.d. uniform A, or A = Ws Hs + 0.05 U with sparse planted factors of rank k "
                         "(smk_matrix_fill_planted: block pivoting keeps exchanging variables for many iterations)")
    ap.add_argument("--single-copy", action="store_true",
                    help="bf16 workloads under MU / HALS: A without its stored transpose (smk_matrix_create_single_copy)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-every-iteration", action="store_true",
                    help="form the stopping rule's metric (gradients for PG_RATIO, W - Wprev for DELTA_FNORM) and read it back after "
                         "EVERY timed iteration, as the reference's default run does past min_iter (nmf_solve_generic.hpp:98-121); "
                         "the default bench line forms no gradients inside the timed region")
    ap.add_argument("--api-path", action="store_true",
                    help="N = 1: time the call every reference caller makes -- Nmf(opts, host fp64 A, W, H) = smk_nmf_dense from a "
                         "host buffer (upload + conversion + transpose + iterations) -- beside the resident number; adds api_path{}")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 without torch.distributed: one process, one host thread per device, communicators from "
                         "ncclCommInitAll (smk_comm_init_all)")
    ap.add_argument("--no-fallback", action="store_true",
                    help="plain `--gpus N` launch: do not try --single-process when the torch.distributed.run ranks fail or stall")
    ap.add_argument("--stall-s", type=float, default=90.0,
                    help="watchdog: a rank that makes no progress for this long (rendezvous, communicator, first collective, a "
                         "timed window) ends the run with a non-zero code; start-up (imports) gets 300 s")
    ap.add_argument("--watchdog-s", type=float, default=1500.0, help="overall wall-clock limit of a launched run")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="MEASUREMENT HOOK, one GPU: run as rank 0 of a world of this many ranks -- its column shard, the chunk "
                         "geometry and the row blocks of that world, every collective issued through RCCL with ONE rank "
                         "(device-local).  Times the per-rank work of an N-GPU run without the xGMI transfers; the factors "
                         "it produces are meaningless (the other ranks' blocks never arrive)")
    return ap.parse_args(argv)


# ---- report ---------------------------------------------------------------------------------------------------------
def build_report(args, world, elapsed, windows, rank0, ranks_report, collectives, parallelism):
    """the ONE JSON line.  rank0: dict with the timers of rank 0's solver (ms0, c0, ms1, c1, bytes, flops)"""
    m, n, k, alg, storage, desc = WORKLOADS[args.workload]
    sharded = world > 1 or args.emulate_world > 1
    avg_ms = (rank0["ms0"] + rank0["ms1"]) / max(rank0["c0"] + rank0["c1"], 1)
    achieved_gbs = rank0["bytes"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    mfma_tf = rank0["flops"] / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
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
        "data": "synthetic" if args.data == "uniform" else "synthetic (planted sparse factors of rank k + 0.05 uniform noise)",
        "config": {"workload": desc + (" [single copy of A: no stored transpose]" if args.single_copy else ""), "m": m, "n": n, "k": k, "algorithm": alg, "A_storage": storage,
                   "state": "W,H,Gram fp64; big products: MFMA with fp32 accumulation folded into fp64",
                   "parallelism": parallelism, "collectives": collectives,
                   "progress_checks": ("after EVERY timed iteration: the stopping rule's metric is formed (gradients / W - Wprev) and read "
                                       "back one iteration late with a snapshot, as smk_solver_run does past min_iter -- the reference's "
                                       "default behaviour (nmf_solve_generic.hpp:98-121)" if args.check_every_iteration else
                                       "none in the timed region; gradients formed on demand (the reference's solvers form gradW / gradH "
                                       "every iteration, nmf_solver_mu.hpp:151-164, nmf_solver_bpp.hpp:370-377: dead work unless the "
                                       "rule is evaluated; --check-every-iteration times the checked loop)")},
        # useful flops (2 k per matrix entry) of the streaming products.  fp32 storage computes them as three fp16
        # MFMAs per product, so its ratio is against the NATIVE fp32 matrix peak that this replaces
        # and may exceed 1; the bound that matters for this path is roofline.frac (HBM).
        "useful_tflops_big_products": mfma_tf,
        "useful_tflops_vs_native_mfma_peak": mfma_tf / MFMA_PEAK_TF[storage],
        "whole_iteration_tflops": 4.0 * m * n * k / (elapsed / args.steps) / 1e12,
        "collectives_ms_per_step_rank0": ranks_report[0]["collectives_ms_per_step"] if sharded else None,
        "exposed_comm_ms_per_step_rank0": ranks_report[0].get("exposed_comm_ms") if sharded else None,
        "per_rank": ranks_report if sharded else None,
        "windows": len(windows), "windows_ms": [round(w * 1e3, 4) for w in windows],
        "timed_region_s": sum(windows),
        "roofline": {
            "bound": "hbm",
            "kernel": ("smk::bigprod_kernel" if storage == "bf16" else "smk::bigprod_f3_kernel (fp32 A as two fp16 terms)")
                      + " (W'A and H*At passes)",
            "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None,
            "avg_launch_ms": avg_ms, "launches": rank0["c0"] + rank0["c1"],
            "pass_WtA_ms": rank0["ms0"] / max(rank0["c0"], 1), "pass_HAt_ms": rank0["ms1"] / max(rank0["c1"], 1),
            "algorithmic_bytes_per_launch": rank0["bytes"],
        },
    }
    # HBM traffic comes from separate rocprofv3 --pmc passes (tools/profile.sh -> profiles/hbm_traffic.json).  The
    # entry records the hash of the kernel source it was measured on: a figure from an older kernel is not printed.
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof) and args.emulate_world <= 1 and not args.single_copy and args.data == "uniform":     # the passes they were taken on
        try:
            pj = json.load(open(prof))
            key = f"{args.workload}_n{world}"
            if key in pj:
                if pj[key].get("kernel_source_sha16") == kernel_source_sha16():
                    out["roofline"]["traffic"] = pj[key]["bytes_per_launch"]
                    out["roofline"]["traffic_source"] = pj[key].get("source", "profiles/")
                else:
                    out["roofline"]["traffic_stale"] = ("profiles/hbm_traffic.json was measured on another version of "
                                                        "smallk_amd/csrc/bigprod.hip; rerun tools/profile.sh")
        except Exception:
            pass
    # MFMA utilisation from counters (north_star asks for it by name): a separate rocprofv3 --pmc pass (tools/pmc_mfma.py ->
    # profiles/mfma_util.json), tied to the kernel source like the traffic figure
    prof = os.path.join(ROOT, "profiles", "mfma_util.json")
    if os.path.exists(prof) and args.emulate_world <= 1 and not args.single_copy:
        try:
            pj = json.load(open(prof))
            key = f"{args.workload}_n{world}"
            if key in pj:
                if pj[key].get("kernel_source_sha16") == kernel_source_sha16():
                    out["roofline"]["mfma_busy_frac"] = pj[key]["mfma_busy_frac"]
                    out["roofline"]["mfma_busy_source"] = pj[key].get("source", "profiles/")
                else:
                    out["roofline"]["mfma_busy_stale"] = "profiles/mfma_util.json was measured on another version of bigprod.hip"
        except Exception:
            pass
    return out


def projection_8gpu(m, n, k, alg, one_gpu_ms):
    """The arithmetic behind DESIGN 7's "about 6.6x at 8 GPUs" for C4 (BPP / MU: row-sharded W update), so that the first SCALE record
    can be read against it.  NOT a measurement: `assumed_bus_GBps` is an assumption about RCCL over 7 xGMI links, `compute_ms` is
    the per-rank time of an 8-rank run measured on ONE GPU (bench.py --emulate-world 8, the newest profiles/rNN_bench_c4_emulate8.json)."""
    import glob
    kpp = 32 * ((k + 31) // 32)
    world, chunks, bus = 8, 4, 300.0
    rs_in = float(m) * kpp * 8                 # (AH')' partial sums, fp64 on the wire, reduce-scattered in `chunks` pieces
    ag = float(m) * k * 4                      # all-gather of the packed operand of W (two fp16 terms per entry)
    small = 2.0 * k * k * 8                    # all-reduces of HH' and W'W
    compute_ms, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_c4_emulate8.json")), reverse=True):
        try:
            compute_ms, src = json.load(open(f))["ms_per_step"], os.path.basename(f)
            break
        except Exception:
            continue
    ms = lambda b: b / (bus * 1e9) * 1e3
    # the chunk pipeline hides every exchange behind a pass except the LAST reduce-scatter chunk, the FIRST all-gather chunk and the
    # two small (latency-bound, ~0.03 ms each) all-reduces
    exposed = ms(rs_in / chunks) + ms(ag / chunks) + 2 * 0.03
    out = {"for": "C4 on 8 GPUs, column-sharded (DESIGN 7): ARITHMETIC, NOT A MEASUREMENT",
           "assumed_bus_GBps": bus, "chunks": chunks,
           "payload_bytes": {"reduce_scatter_of_AHt_in": rs_in, "reduce_scatter_out_per_rank": rs_in / world,
                             "all_gather_of_packed_W": ag, "small_all_reduces": small},
           "transfer_ms_if_nothing_overlapped": ms(rs_in) + ms(ag) + 2 * 0.03,
           "exposed_ms_with_the_chunk_pipeline": exposed,
           "compute_ms": compute_ms, "compute_ms_source": src, "one_gpu_ms": one_gpu_ms}
    if compute_ms and one_gpu_ms:
        out["projected_ms_per_step"] = compute_ms + exposed
        out["projected_speedup_at_8"] = one_gpu_ms / (compute_ms + exposed)
        out["compute_only_speedup_at_8"] = one_gpu_ms / compute_ms
    return out


def api_path_report(args, m, n, k, alg, storage, resident_it_s):
    """The call every reference caller makes (nmf/src/main.cpp:218-233, smallk.cpp:604-619, smallk_lib.pyx:769): Nmf(opts, A, W, H)
    with A a HOST fp64 column-major buffer.  The reference wraps the buffer as a view (common/src/nmf.cpp:224-226); here it crosses
    PCIe once (smk_matrix_upload_f64).  Timed in pieces (create, upload incl. the
    stored transpose, solver set-up + run + factors back) and as the ONE call smk_nmf_dense; matrices that would not leave half of
    the host's free memory are cut to fewer columns (stated)."""
    import numpy as np
    import smallk_amd
    iters = args.warmup + args.steps
    try:
        avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
    except Exception:
        avail = 64 << 30
    n_use = n
    while m * n_use * 8 > 0.4 * avail and n_use > 1024:
        n_use //= 2
    t0 = time.perf_counter()
    G = smallk_amd.DenseMatrix(m, n_use, storage=storage)
    G.fill_uniform(42)
    A = G.download()                     # host fp64 copy of the stored (rounded) values: what a caller would hand over
    G.close()
    t_gen = time.perf_counter() - t0
    W0 = smallk_amd.uniform_host(m, k, 43)
    H0 = smallk_amd.uniform_host(k, n_use, 44) * (2.0 / k)
    opts = smallk_amd.make_options(m, n_use, k, alg, min_iter=iters, max_iter=iters)
    bytes_a = float(m) * n_use * 8.0
    pieces = []
    for rep in range(2):                 # the second pass has the pinned pool and the device allocations warm
        t0 = time.perf_counter()
        M = smallk_amd.DenseMatrix(m, n_use, storage=storage)
        t1 = time.perf_counter()
        M.upload(A)
        t2 = time.perf_counter()
        sv = smallk_amd.NmfSolver(M, opts)
        sv.set_factors(W0, H0)
        rc, its, us = sv.run()
        sv.factors()
        t3 = time.perf_counter()
        sv.close()
        M.close()
        assert rc == 0, rc
        pieces.append({"create_s": t1 - t0, "upload_s": t2 - t1, "upload_GBps": bytes_a / (t2 - t1) / 1e9,
                       "solve_s": t3 - t2, "solver_elapsed_s": us * 1e-6, "iterations": its})
    t0 = time.perf_counter()
    res = smallk_amd.nmf(A, W0, H0, alg, storage=storage, min_iter=iters, max_iter=iters)
    t_call = time.perf_counter() - t0
    assert res.result == 0, res.result
    best = min(pieces, key=lambda p: p["upload_s"])
    return {"what": "smk_nmf_dense(opts, HOST fp64 A, W, H): create + upload (PCIe copy, conversion to the stored type, stored transpose) "
                    "+ solver set-up + iterations + factors back",
            "m": m, "n": n_use, "k": k, "algorithm": alg, "A_storage": storage, "host_bytes_A": bytes_a,
            "columns_cut_to_fit_host_memory": n_use != n,
            "iterations": res.iteration_count, "one_call_s": t_call, "end_to_end_it_s": res.iteration_count / t_call,
            "solver_elapsed_s": res.elapsed_us * 1e-6,
            "upload_GBps": best["upload_GBps"], "upload_s": best["upload_s"],
            "transpose": "inside upload_s (one device pass after the last chunk)",
            "pieces": pieces, "resident_it_s": resident_it_s,
            "upload_in_iterations_at_resident_rate": best["upload_s"] * resident_it_s if n_use == n else None,
            "host_copy_generated_in_s": t_gen,
            "upload": "hipMemcpy2DAsync from the caller's pageable buffer, 64 MB chunks, conversion per chunk (smk_matrix_upload_f64)"}


def per_rank_report(rank, args, k, windows, ms0, c0, ms1, c1, msc, cc, msx=0.0, cx=0, mscal=0.0, ccal=0):
    """per rank and per step: both streaming passes, the collectives' own spans, and what the step spends outside the
    passes.  If the collectives take longer than that remainder, the difference was hidden behind the products:
    overlap_lower_bound = max(0, 1 - outside / collectives)."""
    timed_steps = max(len(windows) * args.steps, 1)
    ngroups = max((k + 63) // 64, 1)
    pr = {"rank": rank, "ms_per_step": sum(windows) / timed_steps * 1e3,
          "products_ms_per_step": (ms0 + ms1) / timed_steps, "collectives_ms_per_step": msc / timed_steps,
          "collective_calls_per_step": cc / timed_steps, "passes_per_step": (c0 + c1) / ngroups / timed_steps}
    # MEASURED exposure: the main stream's waits for events of the collective stream, each bracketed by two events on the main
    # stream (slot 3 of smk_solver_kernel_time); ~5 us per wait is the bracket itself
    # and the same bracket around a wait for an event that completed long ago (slot 4) is what a bracket costs by itself
    null_ms = mscal / ccal if ccal > 0 else 0.0
    pr["exposed_comm_ms_raw"] = msx / timed_steps
    pr["exposed_comm_waits_per_step"] = cx / timed_steps
    pr["wait_bracket_cost_ms"] = null_ms
    pr["exposed_comm_ms"] = max(0.0, (msx - cx * null_ms) / timed_steps)
    pr["outside_products_ms_per_step"] = pr["ms_per_step"] - pr["products_ms_per_step"]
    pr["overlap_lower_bound"] = (max(0.0, 1.0 - pr["outside_products_ms_per_step"] / pr["collectives_ms_per_step"])
                                 if pr["collectives_ms_per_step"] > 0 else None)
    return pr


# ---- one rank per process (N = 1, or under torch.distributed.run) ----------------------------------------------------
def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    _hb["rank"] = rank
    if world > 1 or args.emulate_world > 1:
        start_rank_watchdog(args.stall_s)
    beat("imports", limit=300.0)
    import numpy as np
    import torch
    import smallk_amd
    from smallk_amd import dist as sdist

    if args.emulate_world > 1:
        os.environ["SMK_COMM_FORCE"] = "1"
        os.environ["SMK_COMM_EMULATE_WORLD"] = str(args.emulate_world)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks, or run plain `python bench.py --gpus {args.gpus}`")
    # Test hooks for a 1-GPU box (the multi-rank path otherwise only runs on the driver's 8-GPU node):
    # SMK_BENCH_SHARE_GPU=1 puts every rank on device 0, SMK_BENCH_BACKEND=gloo replaces RCCL (which refuses
    # two ranks on one device) by gloo on device tensors.  Numbers from such a run mean nothing.
    share = os.environ.get("SMK_BENCH_SHARE_GPU", "0") == "1"
    backend = os.environ.get("SMK_BENCH_BACKEND", "nccl")
    device_index = 0 if share else local_rank
    native = backend == "nccl"          # default: RCCL from C (comm.cpp); "gloo": the round-1 callback hook (tests)
    rccl_defaults_set = []              # environment defaults this script added (taken back for plan B)
    if world > 1:
        import torch.distributed as dist
        beat("rendezvous (gloo side channel)", limit=args.stall_s)
        dist.init_process_group("gloo")                 # CPU side channel only: unique id, barrier, max of the clocks
        rccl_env_defaults(rccl_defaults_set)
    beat("device init", limit=args.stall_s)
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    smallk_amd.initialize(device_index)
    comm, fallback_group, collectives = None, None, "none"
    guard = StdoutGuard(world > 1 or args.emulate_world > 1)
    if world > 1 and native:
        # RCCL communicator created by libsmallk_amd.so itself; every collective of the iteration is issued from C
        ok = 1
        try:
            beat("communicator (ncclCommInitRank)", limit=args.stall_s)
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.tensor(list(smallk_amd.Comm.unique_id()), dtype=torch.uint8)
            torch.distributed.broadcast(uid, 0)
            comm = smallk_amd.Comm.init_rank(bytes(uid.tolist()), rank, world)
            beat("communicator self-test (first collectives)", limit=args.stall_s)
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
            beat("torch.distributed nccl group (native communicator failed)", limit=args.stall_s)
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

    total_iters = args.warmup + args.steps
    beat("matrix fill", limit=lim(args, 180.0))
    A = smallk_amd.DenseMatrix(m, n, col0=col0, ncols=ncols, storage=storage, single_copy=args.single_copy)
    if args.data == "planted":
        A.fill_planted(42, k, 0.7, 0.05)
    else:
        A.fill_uniform(42)
    W0 = smallk_amd.uniform_host(m, k, 43)
    # E[A] = 1/2: scale H0 so that W0 H0 has the same mean.  (An unscaled uniform start makes the first HALS
    # W update clamp every entry to zero and the run would iterate on the all-eps guard columns.)
    H0 = smallk_amd.uniform_host(k, ncols, 44, c0=col0, gheight=k) * (2.0 / k)
    opts = smallk_amd.make_options(m, n, k, alg, min_iter=total_iters, max_iter=total_iters)
    beat("solver set-up", limit=lim(args, 180.0))
    solver = smallk_amd.NmfSolver(A, opts)
    if comm is not None:
        solver.attach_comm(comm)
    elif world > 1:
        sdist.attach(solver, rank, world, dev, group=fallback_group)
    solver.set_factors(W0, H0)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    beat("warm-up iterations (first collectives of the solver)", limit=lim(args, 120.0))
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
        if args.check_every_iteration:
            solver.iterate_checked(args.steps)
        else:
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
    beat("window 0", limit=lim(args, 120.0))
    windows = [window()]
    wlimit = max(args.stall_s, 30.0 * windows[0])
    while not test_hook and (len(windows) < 5 or sum(windows) < 0.5) and len(windows) < 200:
        beat(f"window {len(windows)}", limit=wlimit)
        windows.append(window())
    elapsed = sorted(windows)[len(windows) // 2]
    beat("report", limit=lim(args, 400.0))

    dump = os.environ.get("SMK_BENCH_DUMP_W")        # test hook: W (replicated) after the timed steps, rank 0
    if dump:
        Wd, _ = solver.factors(normalize=False)
        if rank == 0:
            np.save(dump, Wd)

    ms0, c0 = solver.kernel_time(0)
    ms1, c1 = solver.kernel_time(1)
    sharded = world > 1 or args.emulate_world > 1
    msc, cc = solver.kernel_time(2) if sharded else (0.0, 0)    # spans of the collectives on the second stream, this rank
    msx, cx = solver.kernel_time(3) if sharded else (0.0, 0)    # the main stream's waits for them (measured exposure)
    mscal, ccal = solver.kernel_time(4) if sharded else (0.0, 0)
    bytes_per_launch, flops_per_launch = solver.kernel_work(0)
    per_rank = per_rank_report(rank, args, k, windows, ms0, c0, ms1, c1, msc, cc, msx, cx, mscal, ccal)
    ranks_report = [per_rank]
    if world > 1:
        ranks_report = [None] * world
        torch.distributed.all_gather_object(ranks_report, per_rank)

    if rank == 0:
        parallelism = (f"column-shard x{world}, one process per GPU" if world > 1 else
                       f"EMULATED rank 0 of {args.emulate_world} on one GPU (not a benchmark line)" if args.emulate_world > 1
                       else "single GPU")
        out = build_report(args, world, elapsed, windows,
                           {"ms0": ms0, "c0": c0, "ms1": ms1, "c1": c1, "bytes": bytes_per_launch, "flops": flops_per_launch},
                           ranks_report, collectives, parallelism)
        if world > 1:
            out["rccl_choices"] = rccl_choices()
        if args.check_every_iteration:
            out["config"]["check_route"] = solver.kernel_name(2)        # where the last check's sums were formed
        if args.workload == "c4" and args.data == "uniform" and not args.single_copy and not args.check_every_iteration:
            out["projection"] = projection_8gpu(m, n, k, alg, out["ms_per_step"] if world == 1 and args.emulate_world <= 1 else None)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, n, k, alg, 1 if storage == "bf16" else 0, data=args.data)
        if args.api_path and world == 1 and args.emulate_world <= 1:
            solver.close()
            A.close()
            out["api_path"] = api_path_report(args, m, n, k, alg, storage, out["value"])
        guard.restore()
        print(json.dumps(out), flush=True)
    if rank == 0 and world > 1:
        rccl_log_excerpt()
    beat("teardown", limit=lim(args, 60.0))
    barrier()
    if comm is not None:
        solver.close()
        comm.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    beat("done")


# ---- one process, one host thread per device (--single-process) -------------------------------------------------------
def run_single_process(args):
    """N column shards on N devices driven by ONE process: a host thread per device with a library context of its own
    (smk_thread_context_begin), communicators from ncclCommInitAll (smk_comm_init_all).  No torch.distributed, no
    rendezvous, no sockets -- a path that shares nothing with the one-process-per-GPU launch except the solver.
    SMK_BENCH_SHARE_GPU=1 (test hook, one GPU): every thread on device 0, the in-process stand-in instead of RCCL."""
    N = args.gpus
    _hb["rank"] = 0
    start_rank_watchdog(args.stall_s)
    beat("imports", limit=300.0)
    import numpy as np
    try:                                   # torch first, as in the one-process-per-GPU path: the process then runs on the same HIP
        import torch  # noqa: F401         # runtime and the same librccl.so (torch's bundled ones) in both plans
    except Exception:
        pass
    import smallk_amd
    from smallk_amd import _lib as L
    from smallk_amd import dist as sdist
    lib = L.lib()
    share = os.environ.get("SMK_BENCH_SHARE_GPU", "0") == "1"
    ndev = lib.smk_device_count()
    if not share and ndev < N:
        raise SystemExit(f"--gpus {N} --single-process: only {ndev} device(s) visible")
    rccl_env_defaults([])
    guard = StdoutGuard(True)
    beat("communicators (ncclCommInitAll)", limit=args.stall_s)
    smallk_amd.initialize(0)
    comms = smallk_amd.Comm.init_local(N) if share else smallk_amd.Comm.init_all(N)
    collectives = ("in-process stand-in, all shards on device 0 (TEST HOOK)" if share else
                   "RCCL from C (comm.cpp), communicators from ncclCommInitAll, one host thread per device")
    m, n, k, alg, storage, desc = WORKLOADS[args.workload]
    total_iters = args.warmup + args.steps
    gate = threading.Barrier(N)
    shared = {"go_on": True, "dts": [0.0] * N, "err": [None] * N, "reports": [None] * N, "rank0": None, "W": None}
    test_hook = bool(os.environ.get("SMK_BENCH_DUMP_W"))
    windows = []

    def wait():
        gate.wait(timeout=lim(args, 600.0))

    def worker(r):
        try:
            smallk_amd.thread_context_begin(0 if share else r)
            comms[r].selftest()             # known sums through the communicator before the run is trusted to it
            col0, ncols = sdist.shard_columns(n, N, r)
            A = smallk_amd.DenseMatrix(m, n, col0=col0, ncols=ncols, storage=storage)
            if args.data == "planted":
                A.fill_planted(42, k, 0.7, 0.05)
            else:
                A.fill_uniform(42)
            W0 = smallk_amd.uniform_host(m, k, 43)
            H0 = smallk_amd.uniform_host(k, ncols, 44, c0=col0, gheight=k) * (2.0 / k)
            opts = smallk_amd.make_options(m, n, k, alg, min_iter=total_iters, max_iter=total_iters)
            solver = smallk_amd.NmfSolver(A, opts)
            solver.attach_comm(comms[r])
            solver.set_factors(W0, H0)
            if r == 0:
                beat("warm-up iterations (first collectives of the solver)", limit=lim(args, 120.0))
            wait()
            solver.iterate(args.warmup)
            rc = solver.sync()
            assert rc == 0, f"solver failed during warm-up: {rc}"
            lib.smk_device_synchronize()
            solver.enable_timing(True)
            while True:
                # EXACTLY args.steps iterations between barrier + device-synchronize pairs; max over the ranks
                if r == 0:
                    beat(f"window {len(windows)}", limit=max(lim(args, 120.0), 30.0 * (windows[0] if windows else 0.0)))
                wait()
                lib.smk_device_synchronize()
                t0 = time.perf_counter()
                solver.iterate(args.steps)
                rcw = solver.sync()
                lib.smk_device_synchronize()
                wait()
                shared["dts"][r] = time.perf_counter() - t0
                assert rcw == 0, f"solver failed: {rcw}"
                wait()
                if r == 0:
                    windows.append(max(shared["dts"]))
                    shared["go_on"] = not test_hook and (len(windows) < 5 or sum(windows) < 0.5) and len(windows) < 200
                wait()
                if not shared["go_on"]:
                    break
            if test_hook:
                Wd, _ = solver.factors(normalize=False)         # every rank takes part in the gather of W
                if r == 0:
                    shared["W"] = Wd
            ms0, c0 = solver.kernel_time(0)
            ms1, c1 = solver.kernel_time(1)
            msc, cc = solver.kernel_time(2)
            msx, cx = solver.kernel_time(3)
            mscal, ccal = solver.kernel_time(4)
            shared["reports"][r] = per_rank_report(r, args, k, windows, ms0, c0, ms1, c1, msc, cc, msx, cx, mscal, ccal)
            if r == 0:
                b, f = solver.kernel_work(0)
                shared["rank0"] = {"ms0": ms0, "c0": c0, "ms1": ms1, "c1": c1, "bytes": b, "flops": f}
            wait()
            solver.close()
            A.close()
            smallk_amd.thread_context_end()
        except BaseException as e:      # a rank that gives up releases the others (they would wait in a collective)
            shared["err"][r] = repr(e)
            print(f"[bench thread {r}] failed: {e!r}", file=sys.stderr, flush=True)
            try:
                gate.abort()
                lib.smk_comm_abort(comms[r]._h)
            except Exception:
                pass

    beat("matrix fill + solver set-up", limit=lim(args, 240.0))
    threads = [threading.Thread(target=worker, args=(r,), name=f"shard-{r}") for r in range(N)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if any(shared["err"]):
        guard.restore()
        print(f"[bench] single-process run failed: {shared['err']}", file=sys.stderr, flush=True)
        rccl_log_excerpt()
        beat("done")
        return 1
    beat("report", limit=lim(args, 60.0))
    elapsed = sorted(windows)[len(windows) // 2]
    if test_hook and shared["W"] is not None:
        np.save(os.environ["SMK_BENCH_DUMP_W"], shared["W"])
    out = build_report(args, N, elapsed, windows, shared["rank0"], shared["reports"], collectives,
                       f"column-shard x{N}, ONE process, one host thread per device")
    out["rccl_choices"] = rccl_choices()
    guard.restore()
    print(json.dumps(out), flush=True)
    if not share:
        rccl_log_excerpt()
    for c in comms:
        c.close()
    beat("done")
    return 0


def main():
    args_list = [a for a in sys.argv[1:] if a != "--in-child"]
    in_child = len(args_list) != len(sys.argv) - 1
    args = parse_args(args_list)
    if args.workload.startswith("s_"):
        import bench_sparse
        sys.exit(bench_sparse.run_sparse(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not in_child:
        # plain invocation: this process starts the ranks and makes no GPU call itself
        sys.exit(launch(args))
    if args.single_process and args.gpus > 1:
        sys.exit(run_single_process(args))
    run_rank(args)


if __name__ == "__main__":
    main()
