#!/bin/bash
# round 3, call n: ranks in (512, 1024]
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_nnls.py -m gpu -x -q -k "above_512 or above_1024 or bad_params or not_positive or above_128" --durations=8 2>&1 | tail -25 > gpurun_out/r03_k1024_tests.txt
python - > gpurun_out/r03_k1024_speed.txt 2>&1 <<'PY'
import numpy as np, time
import smallk_amd; smallk_amd.initialize(0)
from smallk_amd import solver as S
rng = np.random.default_rng(0)
m, n = 16384, 8192
for k in (512, 768, 1024):
    for alg in ("MU", "HALS", "BPP"):
        if alg == "BPP" and k > 768: iters = 1
        else: iters = 3
        W = rng.random((m, k)); H = rng.random((k, n))
        A = (rng.random((m, k)) @ rng.random((k, n))).astype(np.float32)
        t = time.time()
        r = smallk_amd.nmf(A, W, H, alg, min_iter=iters, max_iter=iters, storage="f32")
        dt = time.time() - t
        print(k, alg, "result", r.result, "iters", r.iteration_count, "wall %.3f s" % dt, flush=True)
PY
