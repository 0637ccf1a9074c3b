"""diagnostic: fresh-process runs of the nmf tool on a sparse MatrixMarket file, counting solver failures and distinct outputs"""
import sys, os, subprocess, pathlib, tempfile, hashlib, collections
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import test_reference_callers as T
tmp = pathlib.Path(tempfile.mkdtemp())
d, A, W0, H0, dic = T._data_dir(tmp, 400, 600, 40, 2, "w2.csv", "h2.csv")
NMF = os.path.join("smallk_amd", "bin", "nmf")
N = int(sys.argv[1]); k = sys.argv[2]; alg = sys.argv[3]
extra = sys.argv[4:]
bad = 0; sigs = collections.Counter()
for i in range(N):
    r = subprocess.run([NMF, "--matrixfile", str(d / "reuters.mtx"), "--k", k, "--algorithm", alg, "--miniter", "3", "--maxiter", "3",
                        "--outfile_W", str(tmp / "w.csv"), "--outfile_H", str(tmp / "h.csv"), "--verbose", "0", "--outprecision", "17"] + extra,
                       capture_output=True, text=True)
    if r.returncode != 0 or "failure" in r.stderr:
        bad += 1; print("run", i, r.returncode, r.stderr.strip().splitlines()[-2:], flush=True); continue
    sigs[hashlib.md5(open(tmp / "w.csv", "rb").read() + open(tmp / "h.csv", "rb").read()).hexdigest()[:8]] += 1
print(alg, "k", k, ":", bad, "failures of", N, "distinct outputs", dict(sigs))
