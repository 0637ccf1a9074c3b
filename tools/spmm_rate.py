"""Rate of the sparse gather product by itself (smk_matrix_sparse_product): avg launch time, algorithmic bytes
nnz * (12 + 8 KP) + ncols * 8 KP per launch, GB/s.  usage: python tools/spmm_rate.py [reuters|1m|both] [k ...]
SMK_SPMM_SEG=0 times the round-4 kernel; SMK_SPMM_SEG_LEN / SMK_SPMM_SEG_U tune the segment kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smallk_amd
from smallk_amd.synthetic import term_document, community_graph

which = sys.argv[1] if len(sys.argv) > 1 else "both"
ks = [int(x) for x in sys.argv[2:]] or [16, 32, 64]
smallk_amd.initialize(0)
mats = []
if which in ("reuters", "both"):
    mats.append(("term-document 12411x7984", term_document(12411, 7984, 500_000, seed=1)))
if which in ("1m", "both"):
    mats.append(("community graph 1M x 1M", community_graph(1_000_000, 16, 16, seed=0)[0]))
print(f"# SMK_SPMM_SEG={os.environ.get('SMK_SPMM_SEG', '1')} SEG_LEN={os.environ.get('SMK_SPMM_SEG_LEN', '64')} U={os.environ.get('SMK_SPMM_SEG_U', '8')}")
for name, A in mats:
    m, n = A.shape
    S = smallk_amd.SparseMatrix(A.data, A.indices, A.indptr, A.shape)
    rng = np.random.default_rng(0)
    for k in ks:
        KP = 8 if k <= 8 else 16 if k <= 16 else 32 if k <= 32 else 64 if k <= 64 else 128
        for tr in (False, True):
            X = np.asfortranarray(rng.random((k, n if tr else m)))
            out, ms = S.product(X, transposed=tr, reps=20)
            ref = ((A @ X.T).T if tr else (A.T @ X.T).T)
            err = np.abs(out - ref).max() / np.abs(ref).max()
            nc = m if tr else n
            bytes_ = A.nnz * (12 + 8 * KP) + nc * 8 * KP
            print(f"{name} nnz={A.nnz} k={k} {'(AH^T)^T' if tr else 'W^TA    '}: {ms*1e3:9.1f} us  {bytes_/ms/1e6:8.1f} GB/s  "
                  f"{A.nnz/ms/1e6:7.2f} G entries/s  err {err:.1e}", flush=True)
    S.close()
