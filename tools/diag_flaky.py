"""diagnostic: run the reference example program repeatedly on the synthesised data directory, count solver failures"""
import sys, os, subprocess, pathlib, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import test_reference_callers as T
hold = len(sys.argv) > 2 and sys.argv[2] == "hold"
if hold:
    import smallk_amd
    smallk_amd.initialize(0)          # like the pytest parent: a second process holds a context on the GPU
tmp = pathlib.Path(tempfile.mkdtemp())
d, A, W0, H0, dic = T._data_dir(tmp, 400, 600, 40, 2, "nmf_rank2_init_w.csv", "nmf_rank2_init_h.csv")
run = tmp / "run"; run.mkdir()
bad = 0
N = int(sys.argv[1])
for i in range(N):
    r = subprocess.run([T.EXAMPLE_BIN, str(d)], cwd=run, capture_output=True, text=True, timeout=900)
    if "solver failure" in r.stderr:
        bad += 1
        print("run", i, [l for l in r.stderr.splitlines() if "failure" in l][:2], flush=True)
print("hold" if hold else "alone", ":", bad, "failures of", N)
