#!/bin/bash
# round 3, call z: block pivoting at k in (64, 128] on the tile kernels (SMK_NNLS_TILE128=1) against nnls_bpp_inv128_kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
SMK_NNLS_TILE128=1 python -m pytest tests/test_gpu_nnls.py tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -x -q -k "above_64 or 100 or 128 or 80 or 65 or not_positive or hard or ill" 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/tests.txt
for t in 0 1 0 1; do
  for k in 80 100 128; do SMK_NNLS_TILE128=$t python3 tools/wide_run.py 16384 8192 $k BPP 12 1 2>/dev/null | tail -1 | sed "s/^/tile128=$t /" >> $OUT/times.txt; done
done
SMK_NNLS_TILE128=1 python3 tools/fuzz_parity.py 150 77 > $OUT/fuzz.txt 2>&1
