"""GPU tests of the sharded path on ONE GPU:
 * two column shards driven by two host threads whose all-reduce callback sums the two device
   buffers -- the real sharded device path, checked against the unsharded oracle;
 * the torch.distributed (NCCL = RCCL) glue with a world of one rank."""
import os
import threading

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("alg,storage,quant", [("HALS", "bf16", 1), ("MU", "f32", 0), ("BPP", "f32", 0)])
def test_two_shards_on_one_gpu(gpu, alg, storage, quant):
    import torch
    from smallk_amd import dist as sdist
    m, n, k, iters, world = 1500, 700, 12, 6, 2
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    A = oracle.fill_uniform(m, n, 42, quant=quant)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    ref = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, tol=1e-9)

    barrier = threading.Barrier(world)
    views = [None] * world
    out = [None] * world
    errors = []

    def make_cb(rank, ar):
        def cb(ptr, count, dtype):
            esz = 4 if dtype == 0 else 8
            off = ptr - ar.base
            views[rank] = ar.ws[off:off + count * esz].view(torch.float32 if dtype == 0 else torch.float64)
            barrier.wait(timeout=60)
            if rank == 0:
                total = views[0] + views[1]
                views[0].copy_(total)
                views[1].copy_(total)
                torch.cuda.synchronize()
            barrier.wait(timeout=60)
            return 0
        return cb

    def run(rank):
        try:
            c0, nc = sdist.shard_columns(n, world, rank)
            D = gpu.DenseMatrix(m, n, col0=c0, ncols=nc, storage=storage)
            D.fill_uniform(42)
            s = gpu.NmfSolver(D, gpu.make_options(m, n, k, alg, min_iter=iters, max_iter=iters, tol=1e-9))
            ar = sdist.TorchAllReduce(s.comm_workspace_bytes(), dev)
            s.set_comm(rank, world, make_cb(rank, ar), ar.ptr, ar.nbytes)
            s._keep = ar
            s.set_factors(W0, H0[:, c0:c0 + nc])
            rc, it, _ = s.run()
            W, H = s.factors()
            out[rank] = (rc, it, c0, nc, W, H)
        except Exception as e:     # pragma: no cover
            errors.append(e)
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    H = np.concatenate([o[5] for o in out], axis=1)
    for o in out:
        assert o[0] == 0 and o[1] == iters
        assert rel(o[4], ref.W) < TOL
    assert rel(out[0][4], out[1][4]) < 1e-12          # replicated W identical on both ranks
    assert rel(H, ref.H) < TOL


def test_torch_nccl_world_of_one(gpu):
    """the production glue: torch.distributed all_reduce on views of the comm workspace"""
    import torch
    import torch.distributed as dist
    from smallk_amd import dist as sdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        gpu.set_stream(torch.cuda.current_stream().cuda_stream)
        m, n, k, iters = 2048, 1024, 32, 5
        A = oracle.fill_uniform(m, n, 42, quant=1)
        W0 = oracle.fill_uniform(m, k, 43)
        H0 = oracle.fill_uniform(k, n, 44)
        ref = oracle.nmf(A, W0, H0, "HALS", min_iter=iters, max_iter=iters)
        D = gpu.DenseMatrix(m, n, storage="bf16")
        D.fill_uniform(42)
        s = gpu.NmfSolver(D, gpu.make_options(m, n, k, "HALS", min_iter=iters, max_iter=iters))
        sdist.attach(s, 0, 1, dev)
        s.set_factors(W0, H0)
        rc, it, _ = s.run()
        W, H = s.factors()
        assert rc == 0 and it == iters
        assert rel(W, ref.W) < TOL and rel(H, ref.H) < TOL
    finally:
        dist.destroy_process_group()


def test_bench_two_processes_match_one(tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process
    per rank), squeezed onto this box's single GPU: both ranks on device 0 and gloo on the device
    tensors instead of RCCL (which refuses two ranks on one device).  Everything else is the real
    path: rendezvous, column shards, workspace registration, the all-reduce callback from C, max-over-
    ranks timing, the JSON line.  W after the same number of HALS iterations must match the 1-GPU run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMK_BENCH_SHARE_GPU="1", SMK_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    common = ["--steps", "3", "--warmup", "1", "--workload", "c2", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=root, capture_output=True, text=True,
                        timeout=600, env=dict(env, SMK_BENCH_DUMP_W=str(tmp_path / "w1.npy")))
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2"] + common,
                        cwd=root, capture_output=True, text=True, timeout=600,
                        env=dict(env, SMK_BENCH_DUMP_W=str(tmp_path / "w2.npy")))
    assert r2.returncode == 0, r2.stderr[-2000:]
    line = [l for l in r2.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1                                  # rank 0 prints ONE JSON line
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and out["config"]["parallelism"].startswith("column-shard x2")
    W1, W2 = np.load(tmp_path / "w1.npy"), np.load(tmp_path / "w2.npy")
    assert np.linalg.norm(W1 - W2) / np.linalg.norm(W1) < 1e-5


def _bench_env(tmp_path, name):
    return dict(os.environ, SMK_BENCH_SHARE_GPU="1", SMK_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
                SMK_BENCH_DUMP_W=str(tmp_path / name))


def test_bench_plain_invocation_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run around it (how the driver starts the N = 1 run, and may start
    the others): the process starts two fresh child ranks itself -- it makes no GPU call of its own --, relays rank 0's
    single JSON line and exits 0.  Both ranks on this box's one GPU, gloo instead of RCCL (as the test above)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--workload", "c2", "--no-cpu-baseline"]
    env = _bench_env(tmp_path, "w2.npy")
    env.pop("WORLD_SIZE", None)
    r2 = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--no-fallback"] + common, cwd=root, capture_output=True, text=True,
                        timeout=600, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    line = [l for l in r2.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r2.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    assert len(out["per_rank"]) == 2 and [p["rank"] for p in out["per_rank"]] == [0, 1]
    assert out["launcher"]["attempts"][0]["rc"] == 0 and len(out["launcher"]["attempts"]) == 1
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=root, capture_output=True, text=True,
                        timeout=600, env=_bench_env(tmp_path, "w1.npy"))
    assert r1.returncode == 0, r1.stderr[-2000:]
    W1, W2 = np.load(tmp_path / "w1.npy"), np.load(tmp_path / "w2.npy")
    assert np.linalg.norm(W1 - W2) / np.linalg.norm(W1) < 1e-5


@pytest.mark.parametrize("workload,shards", [("c2", 2), ("c1", 3)])
def test_bench_single_process_mode(tmp_path, workload, shards):
    """`--single-process`: one process, one host thread and one library context per shard, no torch.distributed.  On a
    multi-GPU node the communicators come from ncclCommInitAll; here every shard sits on device 0 and the in-process
    stand-in carries the collectives.  W after K steps = the one-GPU W; also reached as the launcher's plan B."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--workload", workload, "--no-cpu-baseline"]
    env = _bench_env(tmp_path, "ws.npy")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(shards), "--single-process"] + common, cwd=root, capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == shards and len(out["per_rank"]) == shards and out["value"] > 0
    assert "ONE process" in out["config"]["parallelism"]
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=root, capture_output=True, text=True,
                        timeout=600, env=_bench_env(tmp_path, "w1.npy"))
    assert r1.returncode == 0, r1.stderr[-2000:]
    W1, Ws = np.load(tmp_path / "w1.npy"), np.load(tmp_path / "ws.npy")
    assert np.linalg.norm(W1 - Ws) / np.linalg.norm(W1) < 1e-5


def test_bench_falls_back_to_the_single_process_plan(tmp_path):
    """plan A (torch.distributed.run) is made to stall in its first solver collectives; the watchdog ends it and plan B --
    the single-process path -- produces the line.  The whole thing stays under two minutes."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMK_BENCH_SHARE_GPU="1", SMK_BENCH_BACKEND="gloo", SMK_BENCH_TEST_HANG="1:warm-up")
    env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--stall-s", "15", "--steps", "3", "--warmup", "1", "--workload", "c1",
                        "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=600, env=env)
    dt = time.monotonic() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    att = out["launcher"]["attempts"]
    assert len(att) == 2 and att[0]["rc"] != 0 and att[1]["rc"] == 0 and out["n_gpus"] == 2
    assert "WATCHDOG" in r.stderr and dt < 120, dt


def test_bench_hung_rank_ends_nonzero(tmp_path):
    """a rank that stops for good inside the run (TEST HOOK) ends the plain invocation with a non-zero code in well under
    two minutes, with the stage and the Python stacks on stderr"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMK_BENCH_SHARE_GPU="1", SMK_BENCH_BACKEND="gloo", SMK_BENCH_TEST_HANG="1:warm-up")
    env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--no-fallback", "--stall-s", "15", "--steps", "3", "--warmup", "1",
                        "--workload", "c1", "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and time.monotonic() - t0 < 120
    assert "WATCHDOG" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


# ---- native communicators (comm.cpp): collectives issued from C on the solver's streams ------------------------
@pytest.mark.parametrize("alg,storage,quant,k,shards", [("HALS", "bf16", 1, 12, 2), ("MU", "f32", 0, 12, 3), ("BPP", "f32", 0, 12, 2),
                                                        ("BPP", "f32", 0, 40, 3), ("HALS", "f32", 0, 33, 4), ("BPP", "bf16", 1, 64, 2),
                                                        ("BPP", "f32", 0, 140, 2), ("MU", "f32", 0, 200, 3),
                                                        ("HALS", "f32", 0, 100, 2), ("HALS", "bf16", 1, 150, 3)])     # HALS above k = 64: the accurate product form, sharded
def test_sharded_one_shot_matches_single_and_oracle(gpu, alg, storage, quant, k, shards):
    """smk_nmf_dense_sharded with the in-process stand-in for RCCL: `shards` host threads, each with its own
    device context, column shard (uneven: 701 columns), solver and streams, all on this box's one GPU.  The code
    path is the multi-GPU one -- only ncclAllReduce / ncclAllGather are replaced.  Sharded vs unsharded <= 1e-5
    (summation order only), each vs the oracle <= 1e-4, equal iteration counts."""
    m, n, iters = 1500, 701, 6
    A = oracle.fill_uniform(m, n, 42, quant=quant)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    kw = dict(min_iter=iters, max_iter=iters, tol=1e-9)
    ref = oracle.nmf(A, W0, H0, alg, **kw)
    one = gpu.nmf(A, W0, H0, alg, storage=storage, **kw)
    many = gpu.nmf_sharded(A, W0, H0, alg, shards, storage=storage, local_stub=True, **kw)
    assert one.result == many.result == ref.result == 0
    assert one.iteration_count == many.iteration_count == ref.iteration_count == iters
    assert rel(many.W, one.W) < 1e-5 and rel(many.H, one.H) < 1e-5
    assert rel(many.W, ref.W) < TOL and rel(many.H, ref.H) < TOL


@pytest.mark.parametrize("alg,k", [("BPP", 12), ("MU", 12), ("BPP", 40), ("HALS", 12)])
def test_ranks_agree_on_the_product_form_when_one_shard_has_widely_scaled_columns(gpu, alg, k):
    """The accurate product form is selected when the column maxima of A are more than 2^28 apart -- measured per solver, i.e.
    on the rank's LOCAL columns.  Here only the second shard's columns are scaled by 2^+-20: alone it would pick the accurate
    form (other buffers, other collectives: fp64 blocks gathered instead of the packed operand) while rank 0 would not, and the
    mismatched collectives would hang or corrupt the run.  smk_solver_attach_comm agrees the form over the communicator first.
    Also the row-sharded W update UNDER the accurate form (round 4): reduce-scatter, own-block NNLS / MU, fp64 all-gather per
    chunk."""
    m, n, iters = 900, 640, 6
    A = oracle.fill_uniform(m, n, 42)
    A[:, 320::2] *= 2.0 ** 20
    A[:, 321::2] *= 2.0 ** -20
    A = oracle.quantize(A, 0)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    kw = dict(min_iter=iters, max_iter=iters, tol=1e-9)
    ref = oracle.nmf(A, W0, H0, alg, **kw)
    one = gpu.nmf(A, W0, H0, alg, **kw)
    many = gpu.nmf_sharded(A, W0, H0, alg, 2, local_stub=True, **kw)
    assert one.result == many.result == ref.result == 0
    assert rel(many.W, one.W) < 1e-5 and rel(many.H, one.H) < 1e-5
    assert rel(many.W, ref.W) < TOL and rel(many.H, ref.H) < TOL


@pytest.mark.parametrize("m,shards", [(1001, 3), (1001, 7), (1280, 5)])
def test_sharded_bpp_rows_of_w_in_uneven_chunks(gpu, m, shards):
    """BPP: (AH')' is reduce-scattered by row chunks of ceil(m / shards) rows, each rank solves its own rows of W and the
    rows come back by all-gather; the last chunk is short (1001 = 2 x 334 + 333) or the chunks overrun the padded row
    count (5 x 256 = 1280).  The stopping rule sums the W-side projected gradient over the ranks."""
    n, k = 333, 20
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    for kw in (dict(min_iter=5, max_iter=5, tol=1e-9), dict(min_iter=2, max_iter=60, tol=0.05)):
        ref = oracle.nmf(A, W0, H0, "BPP", **kw)
        many = gpu.nmf_sharded(A, W0, H0, "BPP", shards, local_stub=True, **kw)
        assert many.result == ref.result == 0 and many.iteration_count == ref.iteration_count
        assert rel(many.W, ref.W) < TOL and rel(many.H, ref.H) < TOL


@pytest.mark.parametrize("alg", ["MU", "HALS", "BPP"])
def test_sharded_stopping_rule_agrees_on_every_rank(gpu, alg):
    """tolerance-based stop: every rank must leave the loop at the same iteration (the H-side projected-gradient
    sum and the failure flag are all-reduced), which is also the oracle's."""
    import make_golden as mg
    m, n, k = 900, 500, 8
    A = mg.make_A(m, n, k, True, 0)                           # planted low rank: converges in tens of iterations
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    kw = dict(min_iter=3, max_iter=400, tol=0.02)
    ref = oracle.nmf(A, W0, H0, alg, **kw)
    many = gpu.nmf_sharded(A, W0, H0, alg, 3, local_stub=True, **kw)
    assert many.result == ref.result == 0
    assert 3 < ref.iteration_count < 400 and many.iteration_count == ref.iteration_count
    assert rel(many.W, ref.W) < TOL and rel(many.H, ref.H) < TOL


def test_sharded_failure_is_reported_by_all_ranks(gpu):
    """A rank-deficient start makes one NNLS sub-problem non-SPD: every shard returns FAILURE (none hangs)."""
    A = np.ones((64, 40), order="F")
    r = gpu.nmf_sharded(A, np.ones((64, 3)), np.ones((3, 40)), "BPP", 2, local_stub=True, min_iter=1, max_iter=3)
    assert r.result == -4


def test_rccl_world_of_one_native(gpu):
    """The RCCL objects themselves on this box's single GPU: ncclCommInitAll(1) and the unique-id route, attached
    to a solver (ncclAllReduce / ncclAllGather return immediately for one rank inside comm.cpp, communicator
    creation and destruction are real)."""
    from smallk_amd import Comm, NmfSolver, DenseMatrix, make_options
    m, n, k, iters = 1024, 512, 16, 4
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    ref = oracle.nmf(A, W0, H0, "BPP", min_iter=iters, max_iter=iters)
    for make in (lambda: Comm.init_all(1)[0], lambda: Comm.init_rank(Comm.unique_id(), 0, 1)):
        comm = make()
        assert (comm.rank, comm.world) == (0, 1)
        comm.selftest()                 # ncclAllReduce (fp64, fp32) + ncclAllGather really issued, sums checked
        D = DenseMatrix.from_host(A)
        s = NmfSolver(D, make_options(m, n, k, "BPP", min_iter=iters, max_iter=iters))
        s.attach_comm(comm)
        s.set_factors(W0, H0)
        rc, it, _ = s.run()
        W, H = s.factors()
        s.close()
        D.close()
        comm.close()
        assert rc == 0 and it == iters and rel(W, ref.W) < TOL and rel(H, ref.H) < TOL


@pytest.mark.parametrize("alg,storage,quant,k,chunks", [("BPP", "f32", 0, 16, 3), ("BPP", "f32", 0, 64, 2), ("BPP", "bf16", 1, 40, 4),
                                                         ("BPP", "f32", 0, 140, 2), ("HALS", "bf16", 1, 32, 3), ("MU", "f32", 0, 12, 2),
                                                         ("MU", "bf16", 1, 70, 3), ("BPP", "f32", 0, 16, 1)])
def test_rccl_collectives_forced_at_world_one(gpu, monkeypatch, alg, storage, quant, k, chunks):
    """SMK_COMM_FORCE=1: a world of ONE rank runs the whole multi-GPU schedule with the real nccl* calls -- the H*At
    pass in row chunks with ncclReduceScatter (BPP) / ncclAllReduce (MU, HALS) of chunk j on the second stream beside
    the product of chunk j + 1, block-cyclic NNLS of W, W'W from the own blocks + ncclAllReduce, the packed operand
    by ncclAllGather per chunk, W'A accumulated chunk by chunk, the fp64 W gathered at the end.  With one rank every
    collective is the identity, so the result must be the unsharded one; what the test pins is that the RCCL calls,
    the event choreography and the chunk arithmetic are executed and leave the bits alone."""
    from smallk_amd import Comm, NmfSolver, DenseMatrix, make_options
    monkeypatch.setenv("SMK_COMM_FORCE", "1")
    monkeypatch.setenv("SMK_COMM_CHUNKS", str(chunks))
    monkeypatch.setenv("SMK_TIMING_STRIDE", "1")        # count every pass (short passes are otherwise timed one in 16)
    m, n, iters = 2000, 901, 6
    A = oracle.fill_uniform(m, n, 42, quant=quant)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    ref = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters)
    one = gpu.nmf(A, W0, H0, alg, storage=storage, min_iter=iters, max_iter=iters)
    comm = Comm.init_all(1)[0]
    D = DenseMatrix.from_host(A, storage=storage)
    s = NmfSolver(D, make_options(m, n, k, alg, min_iter=iters, max_iter=iters))
    s.attach_comm(comm)
    s.set_factors(W0, H0)
    s.enable_timing(True)
    rc, it, _ = s.run()
    W, H = s.factors()
    ms_c, n_c = s.kernel_time(2)
    _, n_p = s.kernel_time(1)
    s.close()
    D.close()
    comm.close()
    assert rc == 0 and it == iters
    assert n_c > 0 and ms_c >= 0.0                      # collectives were issued (and timed on the second stream)
    assert n_p == (iters + (1 if alg == "HALS" else 0)) * ((k + 63) // 64)       # a pass cut into chunks counts once per group
    assert rel(W, ref.W) < TOL and rel(H, ref.H) < TOL
    assert rel(W, one.W) < (1e-5 if storage == "f32" else 1e-5) and rel(H, one.H) < 1e-5


@pytest.mark.parametrize("chunks,f64", [(1, 0), (3, 0), (2, 1)])
def test_sharded_bpp_chunk_pipeline_on_the_stand_in(gpu, monkeypatch, chunks, f64):
    """3 shards on one device, BPP k = 64 (fp16 two-term products): the chunk count of the exchange and the element type
    of the summed (AH')' (fp64 default, SMK_COMM_F64=0 for fp32) do not change the result beyond summation order.
    (At this size: 3.8e-6 between sharded and unsharded with EITHER element type -- the accumulation order inside the
    streaming products of differently shaped shards.  At the full C4 size on noise-like data the fp32 wire costs 1e-4
    after one iteration, which is why fp64 is the default: tests/test_gpu_fullsize.py.)"""
    monkeypatch.setenv("SMK_COMM_CHUNKS", str(chunks))
    monkeypatch.setenv("SMK_COMM_F64", str(f64))
    m, n, k, iters = 3000, 640, 64, 5
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    kw = dict(min_iter=iters, max_iter=iters, tol=1e-9)
    ref = oracle.nmf(A, W0, H0, "BPP", **kw)
    one = gpu.nmf(A, W0, H0, "BPP", **kw)
    many = gpu.nmf_sharded(A, W0, H0, "BPP", 3, local_stub=True, **kw)
    assert many.result == 0 and many.iteration_count == iters
    assert rel(many.W, ref.W) < TOL and rel(many.H, ref.H) < TOL
    assert rel(many.W, one.W) < 1e-5 and rel(many.H, one.H) < 1e-5


def test_sharded_delta_fnorm_rule_with_row_sharded_w(gpu):
    """BPP with the DELTA_FNORM rule on 2 shards: the rule needs all of W, so the fp64 rows are gathered for every check."""
    from smallk_amd import _lib as L
    import make_golden as mg
    m, n, k = 1100, 480, 6
    A = mg.make_A(m, n, k, True, 0)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44)
    kw = dict(min_iter=2, max_iter=300, tol=0.01, prog_est=L.PROG_DELTA_FNORM)
    ref = oracle.nmf(A, W0, H0, "BPP", **kw)
    many = gpu.nmf_sharded(A, W0, H0, "BPP", 2, local_stub=True, **kw)
    assert many.result == ref.result == 0 and many.iteration_count == ref.iteration_count
    assert rel(many.W, ref.W) < TOL and rel(many.H, ref.H) < TOL


@pytest.mark.parametrize("alg,m", [("BPP", 1501), ("HALS", 1001)])
def test_callback_hook_with_rows_not_divisible_by_world(gpu, alg, m):
    """the callback path (smk_solver_set_comm) with m % world != 0: its workspace layout does not depend on the world"""
    import torch
    from smallk_amd import dist as sdist
    n, k, iters, world = 300, 10, 4, 3
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    ref = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, tol=1e-9)
    barrier = threading.Barrier(world)
    views, out, errors = [None] * world, [None] * world, []

    def make_cb(rank, ar):
        def cb(ptr, count, dtype):
            esz = 4 if dtype == 0 else 8
            off = ptr - ar.base
            views[rank] = ar.ws[off:off + count * esz].view(torch.float32 if dtype == 0 else torch.float64)
            barrier.wait(timeout=60)
            if rank == 0:
                total = views[0] + views[1] + views[2]
                for v in views:
                    v.copy_(total)
                torch.cuda.synchronize()
            barrier.wait(timeout=60)
            return 0
        return cb

    def run(rank):
        try:
            c0, nc = sdist.shard_columns(n, world, rank)
            D = gpu.DenseMatrix(m, n, col0=c0, ncols=nc)
            D.upload(A[:, c0:c0 + nc])
            s = gpu.NmfSolver(D, gpu.make_options(m, n, k, alg, min_iter=iters, max_iter=iters, tol=1e-9))
            ar = sdist.TorchAllReduce(s.comm_workspace_bytes(), dev)       # sized BEFORE the world is known to the solver
            s.set_comm(rank, world, make_cb(rank, ar), ar.ptr, ar.nbytes)
            s._keep = ar
            s.set_factors(W0, H0[:, c0:c0 + nc])
            rc, it, _ = s.run()
            W, H = s.factors()
            out[rank] = (rc, it, W, H)
        except Exception as e:     # pragma: no cover
            errors.append(e)
            try:
                barrier.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors, errors
    H = np.concatenate([o[3] for o in out], axis=1)
    for o in out:
        assert o[0] == 0 and o[1] == iters and rel(o[2], ref.W) < TOL
    assert rel(H, ref.H) < TOL


def test_comm_selftest_on_the_stand_in(gpu):
    """smk_comm_selftest through the in-process communicator: 3 ranks = 3 host threads on this device"""
    import threading
    from smallk_amd import Comm
    comms = Comm.init_local(3)
    errs = []

    def run(c):
        try:
            c.selftest()
        except Exception as e:      # noqa: BLE001
            errs.append(str(e))
    ts = [threading.Thread(target=run, args=(c,)) for c in comms]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for c in comms:
        c.close()
    assert errs == []


def test_nmf_tool_and_facade_shard_over_smk_num_gpus(tmp_path):
    """SMK_NUM_GPUS=N reaches smk_nmf_dense_sharded from the unchanged callers: the `nmf` command line tool (which
    calls Nmf(NmfOptions...), nmf/src/main.cpp:218-233) and smallk::Nmf through the flat API.  On this one-GPU box
    the shards stay on device 0 (SMK_SHARDS_ON_ONE_GPU=1); results must match the single-shard run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(4)
    m, n, k = 300, 221, 6
    A = rng.random((m, n))          # well conditioned: the (AH')' all-reduce travels as fp32, which a nearly
                                    # rank-deficient Gram matrix would amplify beyond the 1e-5 of SURVEY 8e
    np.savetxt(tmp_path / "a.csv", A, delimiter=",")
    np.savetxt(tmp_path / "w0.csv", rng.random((m, k)), delimiter=",")
    np.savetxt(tmp_path / "h0.csv", rng.random((k, n)), delimiter=",")
    tool = os.path.join(root, "smallk_amd", "bin", "nmf")
    outs = {}
    for tag, env in (("one", {}), ("three", {"SMK_NUM_GPUS": "3", "SMK_SHARDS_ON_ONE_GPU": "1"})):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([tool, "--matrixfile", str(tmp_path / "a.csv"), "--k", str(k), "--algorithm", "BPP", "--miniter", "1",
                            "--maxiter", "8", "--tol", "1e-9", "--infile_W", str(tmp_path / "w0.csv"), "--infile_H",
                            str(tmp_path / "h0.csv"), "--outprecision", "15", "--verbose", "0"],
                           cwd=str(d), capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[tag] = (np.loadtxt(d / "w.csv", delimiter=","), np.loadtxt(d / "h.csv", delimiter=","))
    assert rel(outs["three"][0], outs["one"][0]) < 1e-5 and rel(outs["three"][1], outs["one"][1]) < 1e-5
    # smallk::Nmf through the flat handles, in a fresh process (the environment is read per call)
    code = r"""
import sys; sys.path.insert(0, %r)
import numpy as np
from smallk_amd import SmallkAPI
api = SmallkAPI()
api.load_matrix(filepath=%r)
api.nmf(%d, "HALS", infile_W=%r, infile_H=%r, min_iter=1, max_iter=6, tol=1e-9, outdir=%r)
np.save(%r, api.get_W())
""" 
    Ws = {}
    for tag, env in (("one", {}), ("two", {"SMK_NUM_GPUS": "2", "SMK_SHARDS_ON_ONE_GPU": "1"})):
        out = tmp_path / ("api_" + tag)
        out.mkdir()
        src = code % (root, str(tmp_path / "a.csv"), k, str(tmp_path / "w0.csv"), str(tmp_path / "h0.csv"), str(out) + "/",
                      str(out / "W.npy"))
        r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        Ws[tag] = np.load(out / "W.npy")
    assert rel(Ws["two"], Ws["one"]) < 1e-5


# ---- two or more real GPUs (skipped on the one-GPU test boxes; the driver's 8-GPU node runs them) -------------------
def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:      # pragma: no cover
        return 0


needs_two_gpus = pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs")


@needs_two_gpus
@pytest.mark.parametrize("alg,k", [("BPP", 16), ("BPP", 64), ("HALS", 32), ("MU", 12)])
def test_rccl_two_devices_match_one(gpu, alg, k):
    """smk_nmf_dense_sharded over real devices: ncclCommInitAll, one host thread per device, every collective of the chunk
    pipeline over xGMI.  Against the single-GPU run (<= 1e-5) and the oracle (<= 1e-4)."""
    shards = min(_device_count(), 4)
    m, n, iters = 3000, 1201, 6
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    kw = dict(min_iter=iters, max_iter=iters, tol=1e-9)
    ref = oracle.nmf(A, W0, H0, alg, **kw)
    one = gpu.nmf(A, W0, H0, alg, **kw)
    many = gpu.nmf_sharded(A, W0, H0, alg, shards, local_stub=False, **kw)
    assert many.result == 0 and many.iteration_count == iters
    assert rel(many.W, one.W) < 1e-5 and rel(many.H, one.H) < 1e-5
    assert rel(many.W, ref.W) < TOL and rel(many.H, ref.H) < TOL


@needs_two_gpus
def test_bench_two_ranks_over_rccl(tmp_path):
    """bench.py exactly as the driver launches it for N = 2: one process per GPU, the native RCCL communicator."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--workload", "c2", "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=root, capture_output=True, text=True,
                        timeout=600, env=dict(env, SMK_BENCH_DUMP_W=str(tmp_path / "w1.npy")))
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2"] + common,
                        cwd=root, capture_output=True, text=True, timeout=900,
                        env=dict(env, SMK_BENCH_DUMP_W=str(tmp_path / "w2.npy")))
    assert r2.returncode == 0, r2.stderr[-3000:]
    line = [l for l in r2.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and "RCCL from C" in out["config"]["collectives"] and len(out["per_rank"]) == 2
    W1, W2 = np.load(tmp_path / "w1.npy"), np.load(tmp_path / "w2.npy")
    assert np.linalg.norm(W1 - W2) / np.linalg.norm(W1) < 1e-5


def test_matrix_follows_a_stream_set_after_its_creation(gpu):
    """initialize(), DenseMatrix(...), set_stream(torch's stream), NmfSolver(D): the matrix was created under the
    library's own stream, which set_stream destroys -- the solver must run on the new one."""
    import torch
    m, n, k, iters = 700, 300, 8, 4
    A = oracle.fill_uniform(m, n, 42)
    W0 = oracle.fill_uniform(m, k, 43)
    H0 = oracle.fill_uniform(k, n, 44) * (2.0 / k)
    ref = oracle.nmf(A, W0, H0, "BPP", min_iter=iters, max_iter=iters)
    gpu.finalize()
    gpu.initialize(0)                       # fresh context with its own stream
    D = gpu.DenseMatrix.from_host(A)
    side = torch.cuda.Stream()
    gpu.set_stream(side.cuda_stream)        # destroys the stream D was created under
    s = gpu.NmfSolver(D, gpu.make_options(m, n, k, "BPP", min_iter=iters, max_iter=iters))
    s.set_factors(W0, H0)
    rc, it, _ = s.run()
    W, H = s.factors()
    s.close()
    D.close()
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    assert rc == 0 and it == iters and rel(W, ref.W) < TOL and rel(H, ref.H) < TOL


@pytest.mark.parametrize("workload", ["c4s2", "c3s2"])
def test_bench_two_ranks_at_the_shard_geometry_of_the_8_gpu_run(tmp_path, workload):
    """First-real-run readiness: bench.py under torch.distributed.run with two ranks, each holding exactly the
    shard an 8-GPU run of C4 (262144 x 8192, k = 64, BPP) / C3 (65536 x 2048, k = 32, HALS) gives a rank -- both ranks on this
    box's one GPU, gloo instead of RCCL.  The JSON line must carry what the first run on a node will be read by: per-rank
    products / collectives / exposure, the rccl_choices key, and numbers that add up."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMK_BENCH_SHARE_GPU="1", SMK_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("SMK_BENCH_DUMP_W", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload,
                        "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "rccl_choices" in out
    assert len(out["per_rank"]) == 2 and [p["rank"] for p in out["per_rank"]] == [0, 1]
    for p in out["per_rank"]:
        for key in ("ms_per_step", "products_ms_per_step", "collectives_ms_per_step", "exposed_comm_ms", "exposed_comm_ms_raw",
                    "wait_bracket_cost_ms", "outside_products_ms_per_step", "passes_per_step"):
            assert key in p, key
        assert 0.0 < p["products_ms_per_step"] <= p["ms_per_step"] * 1.05                   # the passes are part of the step
        assert abs(p["outside_products_ms_per_step"] - (p["ms_per_step"] - p["products_ms_per_step"])) < 1e-9
        assert p["exposed_comm_ms"] >= 0.0 and p["exposed_comm_ms"] <= p["ms_per_step"]
        assert abs(p["passes_per_step"] - 2.0) < 0.01
    # whole-job rate = steps / the MEDIAN window (each window = the max over ranks between barrier + synchronize pairs)
    w = sorted(out["windows_ms"])
    assert abs(out["value"] - out["steps"] / (w[len(w) // 2] * 1e-3)) <= 1e-3 * out["value"]          # (windows_ms is rounded to 0.1 us)
    assert abs(out["ms_per_step"] - w[len(w) // 2] / out["steps"]) <= 1e-3 * out["ms_per_step"]
