"""bench_sparse.py -- the sparse-A workloads of bench.py (`--workload s_reuters | s_reuters_hals | s_1m`), one GPU.

The reference's only published timings are sparse NMF on its Reuters term-document matrix (12411 x 7984,
sphinx/source/pages_smallkAPI.rst:58-92: BPP k = 32, 39 iterations in 4.354 s = 9.0 it/s; :110-143: HALS k = 16, 88 iterations
in 1.560 s = 56 it/s; hardware not stated, 8 threads).  That file (smallk_data/reuters.mtx, Makefile:28) is not in the tree:
`s_reuters*` run on a SYNTHETIC term-document matrix of the same shape (smallk_amd/synthetic.py:term_document, ~0.47 M stored
entries = 0.48 % dense, Zipf term frequencies), so the numbers stand beside the reference's, not against them.  `s_1m` is
C5's matrix (10^6-node community graph, 16 M entries) under BPP at k = 32.

A step = one NMF iteration: the two gather products (spmm_seg.hip / spmm_gather) + Gram matrices + the factor updates.
roofline (round 6): the gather product is bound by the rate at which 8 KP-byte factor rows can be GATHERED, not by HBM -- the factor
(256 MB at 10^6 x 32) is served by L2 / the Infinity Cache.  `bound: "gather"`, algorithmic bytes per launch = nnz (12 + 8 KP)
[value + row index + one gathered factor row per stored entry] + ncols 8 KP [the result]; `peak` = the rate a pure row-gather
kernel reaches on this chip for rows of that size from a table of the factor's size (tools/mb/mb_gather.hip gatherrow<RB>,
profiles/r06_row_gather_ceiling.txt), plus the stream of values and indices at the same time.  Beside it `hbm_min_bytes` =
nnz 12 + (rows + cols) 8 KP (every array touched once) with its own fraction of the 8 TB/s HBM peak, the share of the step the
block-pivoting launches take, and the counter-based `traffic` (profiles/hbm_traffic.json, key s_1m_n1)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

SPARSE_WORKLOADS = {
    # name: (generator, m, n, nnz target, k, algorithm, description)
    "s_reuters": ("term_document", 12411, 7984, 500_000, 32, "BPP",
                  "sparse 12411x7984 synthetic term-document (Reuters shape), k=32 BPP"),
    "s_reuters_hals": ("term_document", 12411, 7984, 500_000, 16, "HALS",
                       "sparse 12411x7984 synthetic term-document (Reuters shape), k=16 HALS"),
    "s_1m": ("community_graph", 1_000_000, 1_000_000, 16_000_000, 32, "BPP",
             "sparse 10^6 x 10^6 community graph (C5's matrix), 16 M entries, k=32 BPP"),
}
HBM_PEAK_GBS = 8000.0
# measured ceiling of a pure row gather (tools/mb/mb_gather.hip, profiles/r06_row_gather_ceiling.txt): TB/s of gathered bytes for
# rows of 8 KP bytes -- {row bytes: (table in L2, table of 256 MB = Infinity Cache, table of 1 GB = HBM)}; filled from the profile
# file when it is present (bench lines say which figure they used)
GATHER_CEILING_FILE = os.path.join(ROOT, "profiles", "r06_row_gather_ceiling.json")


def gather_peak(row_bytes, table_bytes):
    """GB/s of gathered bytes a pure row-gather reaches for rows of `row_bytes` from a table of `table_bytes` (measured; None
    when the profile has not been taken)"""
    try:
        j = json.load(open(GATHER_CEILING_FILE))
    except Exception:
        return None, None
    rows = j.get(str(int(row_bytes)))
    if not rows:
        return None, None
    # rows: list of [table MB, GB/s]; take the smallest table that holds ours
    rows = sorted(rows)
    for mb, gbs in rows:
        if table_bytes <= mb * (1 << 20):
            return gbs, f"gatherrow{int(row_bytes)} from a {mb} MB table"
    return rows[-1][1], f"gatherrow{int(row_bytes)} from a {rows[-1][0]} MB table"


def make_matrix(name):
    from smallk_amd import synthetic
    gen, m, n, nnz, k, alg, desc = SPARSE_WORKLOADS[name]
    if gen == "term_document":
        return synthetic.term_document(m, n, nnz, seed=1)
    return synthetic.community_graph(m, nnz // m, 16, seed=0)[0]


def cpu_baseline_sparse(name, A, k, alg, budget_s=20.0):
    """the oracle's sparse path (orc_nmf_sparse, C/OpenMP fp64) on the same matrix -- the whole matrix for the Reuters shape,
    a 100 000-node graph of the same degree for s_1m (priced by stored entries)"""
    import numpy as np
    import oracle
    scale = 1.0
    sample = "the same matrix"
    if A.shape[0] > 200_000:
        from smallk_amd import synthetic
        As = synthetic.community_graph(100_000, 16, 16, seed=0)[0]
        scale = A.nnz / As.nnz
        sample = f"a 100000-node graph of the same generator ({As.nnz} entries), priced x{scale:.2f} by stored entries"
        A = As
    m, n = A.shape
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    oracle.nmf_sparse(A, W0, H0, alg, min_iter=1, max_iter=1)
    iters, t, r = 2, 0.0, None
    while True:
        t0 = time.perf_counter()
        r = oracle.nmf_sparse(A, W0, H0, alg, min_iter=iters, max_iter=iters)
        t = time.perf_counter() - t0
        if t > budget_s / 2 or iters >= 64:
            break
        iters *= 2
    t_iter = t / r.iteration_count * scale
    return {"value": 1.0 / t_iter, "unit": "iterations/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"oracle sparse path (C/OpenMP fp64 restatement of NmfSparse) on {sample}: {r.iteration_count} iterations "
                      f"in {t:.2f} s"}


def run_sparse(args):
    import numpy as np
    import torch
    import smallk_amd
    name = args.workload
    gen, m, n, nnz_t, k, alg, desc = SPARSE_WORKLOADS[name]
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("the sparse workloads are one-GPU measurements (replicas only: no bench line for N > 1)")
    torch.cuda.set_device(0)
    smallk_amd.initialize(0)
    t0 = time.perf_counter()
    A = make_matrix(name)
    t_gen = time.perf_counter() - t0
    S = smallk_amd.SparseMatrix(A.data, A.indices, A.indptr, A.shape)
    total = args.warmup + args.steps
    opts = smallk_amd.make_options(m, n, k, alg, min_iter=total, max_iter=total)
    solver = smallk_amd.NmfSolver(S, opts)
    W0 = smallk_amd.uniform_host(m, k, 43)
    H0 = smallk_amd.uniform_host(k, n, 44)
    solver.set_factors(W0, H0)
    solver.iterate(args.warmup)
    assert solver.sync() == 0
    torch.cuda.synchronize()
    solver.enable_timing(True)

    def window():
        torch.cuda.synchronize()
        t = time.perf_counter()
        if getattr(args, "check_every_iteration", False):
            solver.iterate_checked(args.steps)
        else:
            solver.iterate(args.steps)
        rc = solver.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        assert rc == 0, rc
        return dt

    windows = [window()]
    while (len(windows) < 5 or sum(windows) < 0.5) and len(windows) < 200:
        windows.append(window())
    elapsed = sorted(windows)[len(windows) // 2]
    ms0, c0 = solver.kernel_time(0)
    ms1, c1 = solver.kernel_time(1)
    msn, cn = solver.kernel_time(5) if alg == "BPP" else (0.0, 0)        # the block-pivoting launches (both sides)
    KP = 8 if k <= 8 else 16 if k <= 16 else 32 if k <= 32 else 64 if k <= 64 else 128
    bytes0 = A.nnz * (12.0 + 8.0 * KP) + n * 8.0 * KP
    bytes1 = A.nnz * (12.0 + 8.0 * KP) + m * 8.0 * KP
    avg_ms = (ms0 + ms1) / max(c0 + c1, 1)
    achieved = 0.5 * (bytes0 + bytes1) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    kernels = sorted({solver.kernel_name(0), solver.kernel_name(1)})
    check_route = solver.kernel_name(2) if getattr(args, "check_every_iteration", False) else None
    hbm_min = A.nnz * 12.0 + (m + n) * 8.0 * KP                 # every array touched once per launch (average of the two passes)
    peak_g, peak_src = gather_peak(8 * KP, max(m, n) * 8 * KP)
    # the stream of values + indices rides along at 12 B per gathered row: scale the pure-gather ceiling to algorithmic bytes
    peak = peak_g * (12.0 + 8.0 * KP) / (8.0 * KP) if peak_g else None
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        traffic = tj.get(f"{name}_n1", {}).get("bytes_per_launch")
    except Exception:
        pass
    out = {
        "metric": "NMF iterations/sec", "value": args.steps / elapsed, "unit": "iterations/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": desc, "m": m, "n": n, "nnz": int(A.nnz), "k": k, "algorithm": alg,
                   "state": "CSC values, W, H, Gram matrices and every product fp64",
                   "progress_checks": ("after EVERY timed iteration (smk_solver_iterate_checked)" if getattr(args, "check_every_iteration", False)
                                       else "none in the timed region; gradients formed on demand"),
                   "check_route": check_route,
                   "generator": f"smallk_amd/synthetic.py:{gen}", "generate_s": round(t_gen, 2),
                   "longest_column": int(np.diff(A.indptr).max()), "longest_row": int(np.diff(A.tocsr().indptr).max())},
        "windows": len(windows), "windows_ms": [round(w * 1e3, 4) for w in windows], "timed_region_s": sum(windows),
        "roofline": {"bound": "gather", "kernel": " + ".join(kernels),
                     "achieved": achieved, "peak": peak if peak else HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / (peak if peak else HBM_PEAK_GBS),
                     "peak_source": (f"measured pure row gather ({peak_src}), scaled by (12 + row) / row for the value + index stream"
                                     if peak else "no gather ceiling on file: HBM peak"),
                     "traffic": traffic, "avg_launch_ms": avg_ms, "launches": c0 + c1,
                     "pass_WtA_ms": ms0 / max(c0, 1), "pass_HAt_ms": ms1 / max(c1, 1),
                     "algorithmic_bytes_per_launch": 0.5 * (bytes0 + bytes1),
                     "hbm_min_bytes": hbm_min, "hbm_min_frac": hbm_min / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_ms > 0 else 0.0,
                     "products_share_of_step": (ms0 + ms1) / max(len(windows) * args.steps, 1) / (elapsed / args.steps * 1e3),
                     "nnls_share_of_step": msn / max(len(windows) * args.steps, 1) / (elapsed / args.steps * 1e3),
                     "nnls_avg_launch_ms": msn / max(cn, 1)},
        "reference_published": {"s_reuters": "BPP k=32: 39 iterations in 4.354 s = 9.0 it/s (pages_smallkAPI.rst:58-92; its own "
                                             "reuters.mtx, hardware not stated)",
                                "s_reuters_hals": "HALS k=16: 88 iterations in 1.560 s = 56 it/s (pages_smallkAPI.rst:110-143)",
                                }.get(name),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_sparse(name, A, k, alg)
    print(json.dumps(out), flush=True)
    solver.close()
    S.close()
    return 0
