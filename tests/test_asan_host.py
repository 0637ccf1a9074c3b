"""Host C++ of the product (facade.cpp, hierclust.cpp, flatclust.cpp, the three command line tools: ~4000 lines)
under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU.  tests/asan/stub_device.cpp replaces the device
half (solver.cpp + kernels) and delegates every factorisation to the CPU oracle, so what runs here is exactly the
host logic: option handling, file parsing, buffer management, the HierNMF2 tree search, the result writers.
(GPU sanitizers are not available on the pool; SURVEY section 5 asks for this CPU target.)"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN = os.path.join(ROOT, "tests", "asan")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
           OMP_NUM_THREADS="2")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    r = subprocess.run(["make", "-C", ASAN, "-s", "-j4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return os.path.join(ASAN, "_build")


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    import scipy.sparse as sp
    d = tmp_path_factory.mktemp("asan_data")
    rng = np.random.default_rng(0)
    m, n, k = 40, 30, 3
    Wt, Ht = rng.random((m, 2)), rng.random((2, n))
    A = Wt @ Ht + 0.01 * rng.random((m, n))
    np.savetxt(d / "a.csv", A, delimiter=",")
    np.savetxt(d / "w_init.csv", rng.random((m, k)), delimiter=",")
    np.savetxt(d / "h_init.csv", rng.random((k, n)), delimiter=",")
    np.savetxt(d / "w_bad.csv", rng.random((m - 7, k + 1)), delimiter=",")            # wrong shape on purpose
    S = sp.random(60, 45, density=0.25, random_state=1, format="coo", data_rvs=lambda s: rng.random(s) + 0.1)
    S = (S + sp.eye(60, 45) * 0.05).tocoo()
    with open(d / "a.mtx", "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (S.shape[0], S.shape[1], S.nnz))
        for r, c, v in zip(S.row, S.col, S.data):
            f.write("%d %d %.17g\n" % (r + 1, c + 1, v))
    (d / "dictionary.txt").write_text("".join("term%d\n" % i for i in range(60)))
    for i in range(1, 40):                                                            # --initdir files for hierclust
        np.savetxt(d / ("Winit_%d.csv" % i), rng.random((60, 2)), delimiter=",")
        np.savetxt(d / ("Hinit_%d.csv" % i), rng.random((2, 45)), delimiter=",")
    return d


def run(cmd, cwd, ok_codes=(0,)):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600, env=ENV)
    text = r.stdout + r.stderr
    assert "AddressSanitizer" not in text and "runtime error:" not in text and "LeakSanitizer" not in text, text[-4000:]
    assert r.returncode in ok_codes, text[-3000:]
    return r


def test_public_api_tour_is_clean(built, data):
    r = run([os.path.join(built, "api_tour"), str(data)], cwd=str(data))
    assert "api_tour: 0 failures" in r.stdout


@pytest.mark.parametrize("alg", ["MU", "HALS", "BPP", "RANK2"])
def test_nmf_tool_is_clean(built, data, alg):
    k = "2" if alg == "RANK2" else "3"
    run([os.path.join(built, "nmf"), "--matrixfile", str(data / "a.csv"), "--k", k, "--algorithm", alg, "--miniter", "1",
         "--maxiter", "5", "--outprecision", "8", "--verbose", "0"], cwd=str(data))
    run([os.path.join(built, "nmf"), "--matrixfile", str(data / "a.mtx"), "--k", k, "--algorithm", alg, "--maxiter", "4",
         "--verbose", "0"], cwd=str(data))


def test_nmf_tool_error_paths_are_clean(built, data):
    nmf = os.path.join(built, "nmf")
    codes = tuple(range(0, 256))
    run([nmf], cwd=str(data), ok_codes=codes)                                                        # usage
    run([nmf, "--matrixfile", str(data / "missing.csv"), "--k", "3"], cwd=str(data), ok_codes=codes)
    run([nmf, "--matrixfile", str(data / "a.csv"), "--k", "0"], cwd=str(data), ok_codes=codes)
    run([nmf, "--matrixfile", str(data / "a.csv"), "--k", "3", "--infile_W", str(data / "w_bad.csv"), "--infile_H",
         str(data / "h_init.csv")], cwd=str(data), ok_codes=codes)
    run([nmf, "--matrixfile", str(data / "a.csv"), "--k", "3", "--tol", "7"], cwd=str(data), ok_codes=codes)
    run([nmf, "--matrixfile", str(data / "a.csv"), "--k", "3", "--algorithm", "nonsense"], cwd=str(data), ok_codes=codes)


@pytest.mark.parametrize("fmt", ["XML", "JSON"])
def test_hierclust_tool_is_clean(built, data, fmt):
    hc = os.path.join(built, "hierclust")
    common = ["--matrixfile", str(data / "a.mtx"), "--dictfile", str(data / "dictionary.txt"), "--format", fmt, "--verbose", "0"]
    run([hc] + common + ["--clusters", "4", "--maxterms", "3"], cwd=str(data))
    run([hc] + common + ["--clusters", "3", "--initdir", str(data) + "/", "--assignfile", "asg.csv", "--treefile", "tree.out"],
        cwd=str(data), ok_codes=tuple(range(0, 256)))
    run([hc] + common + ["--clusters", "3", "--flat", "1"], cwd=str(data), ok_codes=tuple(range(0, 256)))
    run([hc, "--matrixfile", str(data / "a.mtx")], cwd=str(data), ok_codes=tuple(range(0, 256)))     # missing arguments


@pytest.mark.parametrize("alg", ["HALS", "BPP", "RANK2"])
def test_flatclust_tool_is_clean(built, data, alg):
    fc = os.path.join(built, "flatclust")
    k = "2" if alg == "RANK2" else "3"
    run([fc, "--matrixfile", str(data / "a.mtx"), "--dictfile", str(data / "dictionary.txt"), "--clusters", k, "--algorithm", alg,
         "--maxiter", "6", "--verbose", "0", "--fuzzyfile", "fuzzy.csv"], cwd=str(data), ok_codes=tuple(range(0, 256)))


@pytest.mark.parametrize("init", ["seeded", "initdir"])
def test_hierclust_two_device_step_gives_the_one_device_tree(built, data, init):
    """SMK_CLUST_DEVICES=2: the second TrialSplit of every step runs on a worker thread with its own context and copy of A
    (here: a second host thread over the stub device), speculating on the first child's share of the initialisers.  The
    tree and the assignment files must be byte-identical to the one-device run -- under ASAN + UBSan.  (One OpenMP
    thread: the stub's factorisations come from the CPU oracle, whose reductions change their summation order with the
    team size, and two concurrent callers do not always get the same team.)"""
    hc = os.path.join(built, "hierclust")
    common = [hc, "--matrixfile", str(data / "a.mtx"), "--dictfile", str(data / "dictionary.txt"), "--format", "JSON",
              "--verbose", "0", "--clusters", "5", "--maxterms", "3", "--seed", "11"]
    if init == "initdir":
        common += ["--initdir", str(data) + "/"]
    outs = {}
    runs = (("one", {}), ("two", {"SMK_CLUST_DEVICES": "2", "SMK_SHARDS_ON_ONE_GPU": "1"}),
            ("four", {"SMK_CLUST_DEVICES": "4", "SMK_SHARDS_ON_ONE_GPU": "1"}),          # + one speculative step ahead (round 4)
            ("eight", {"SMK_CLUST_DEVICES": "8", "SMK_SHARDS_ON_ONE_GPU": "1"}))         # + three
    for tag, extra in runs:
        d = data / f"hier_{init}_{tag}"
        d.mkdir()
        r = subprocess.run(common + ["--assignfile", "asg.csv", "--treefile", "tree.json"], cwd=str(d), capture_output=True,
                           text=True, timeout=600, env=dict(ENV, OMP_NUM_THREADS="1", **extra))
        text = r.stdout + r.stderr
        assert "AddressSanitizer" not in text and "runtime error:" not in text and "LeakSanitizer" not in text, text[-4000:]
        assert r.returncode == 0, text[-3000:]
        outs[tag] = ((d / "asg.csv").read_bytes(), (d / "tree.json").read_bytes())
    assert outs["one"] == outs["two"] == outs["four"] == outs["eight"]
