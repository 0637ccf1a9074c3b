"""How far the product-form error drifts over LONG runs at k <= 64 (the reference's default is max_iter = 5000): the distance between
the device result and the oracle after 50 / 100 / 200 / 500 iterations on data with sparse planted factors, default product forms
(fp16 two-term for MU / BPP, bf16x3 for HALS), the accurate form (SMK_NSPLIT=8) and block pivoting under the run-time guard
(SMK_GUARD_EVERY=10), each in its own process.
  python3 tools/long_runs_500.py [child]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

CASES = [("BPP", 1500, 1100, 64), ("BPP", 1500, 1100, 32), ("HALS", 1200, 1000, 64), ("HALS", 1200, 1000, 32), ("MU", 1200, 1000, 64)]
CHECKPOINTS = (50, 100, 200, 500)
if os.environ.get("SMK_LONG_CASE"):              # e.g. SMK_LONG_CASE=BPP,8192,4096,64 SMK_LONG_AT=50,100
    a, m_, n_, k_ = os.environ["SMK_LONG_CASE"].split(",")
    CASES = [(a, int(m_), int(n_), int(k_))]
if os.environ.get("SMK_LONG_AT"):
    CHECKPOINTS = tuple(int(x) for x in os.environ["SMK_LONG_AT"].split(","))

def child():
    import oracle, smallk_amd
    smallk_amd.initialize(0); oracle.set_num_threads(16)
    for alg, m, n, k in CASES:
        rng = np.random.default_rng(17 + k)
        r = k + 2
        A = (rng.random((m, r)) * (rng.random((m, r)) > 0.7)) @ (rng.random((r, n)) * (rng.random((r, n)) > 0.7)) + 0.05 * rng.random((m, n))
        A = oracle.quantize(A, 0)
        W0, H0 = oracle.fill_uniform(m, k, 21), oracle.fill_uniform(k, n, 22)
        if os.environ.get("SMK_LEG_BPP_ONLY") and alg != "BPP":
            continue
        out = []
        for iters in CHECKPOINTS:
            ref = oracle.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, tol=1e-14)
            got = smallk_amd.nmf(A, W0, H0, alg, min_iter=iters, max_iter=iters, tol=1e-14)
            ew = np.linalg.norm(got.W - ref.W) / np.linalg.norm(ref.W); eh = np.linalg.norm(got.H - ref.H) / np.linalg.norm(ref.H)
            out.append(f"{iters}: {max(ew, eh):.1e}")
        label = os.environ.get("SMK_LEG", "default form")
        print(f"{label:>15}  {alg:4s} {m}x{n} k={k}: max(relW, relH) after " + "  ".join(out), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        legs = [("default form", {}),                               # BPP at k > 32 on these small matrices: the accurate form
                ("fp16 form", {"SMK_BPP_SMALL_ACCURATE": "0"}),     # ... and what larger matrices take (8 .. 1 stages per fold by length)
                ("accurate form", {"SMK_NSPLIT": "8"}),
                ("guard every 10", {"SMK_GUARD_EVERY": "10", "SMK_LEG_BPP_ONLY": "1", "SMK_BPP_SMALL_ACCURATE": "0"}),
                ("fold every 4", {"SMK_BP_VARIANT": "108", "SMK_LEG_BPP_ONLY": "1", "SMK_BPP_SMALL_ACCURATE": "0"}),       # fp32 accumulators folded into fp64 twice as often
                ("fold every 2", {"SMK_BP_VARIANT": "128", "SMK_LEG_BPP_ONLY": "1", "SMK_BPP_SMALL_ACCURATE": "0"}),
                ("fold every 1", {"SMK_BP_VARIANT": "129", "SMK_LEG_BPP_ONLY": "1", "SMK_BPP_SMALL_ACCURATE": "0"}),
                ("fold every 8", {"SMK_BP_VARIANT": "125", "SMK_LEG_BPP_ONLY": "1", "SMK_BPP_SMALL_ACCURATE": "0"}),
                ("bf16x3", {"SMK_NSPLIT": "3", "SMK_LEG_BPP_ONLY": "1"})]
        if len(sys.argv) > 1:
            legs = [l for l in legs if l[0] in sys.argv[1:]]
        for name, env in legs:
            env = dict(env, SMK_LEG=name)
            p = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
            print("\n".join(l for l in p.stdout.splitlines() if "max(relW" in l), flush=True)
            if p.returncode:
                print(p.stderr[-1500:])
