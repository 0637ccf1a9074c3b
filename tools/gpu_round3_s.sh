#!/bin/bash
# round 3, call s: the inverse of the Gram matrix above k = 128 on the side stream (tests, fuzz, timing)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03s; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_nnls.py tests/test_gpu_parity.py tests/test_gpu_flatclust.py tests/test_gpu_dist.py tests/test_sparse.py -m gpu -x -q -k "above or not_positive or ill_cond or hard or nnls_hals or wide or 200 or 150 or 129" 2>&1 | tail -5 > $OUT/tests.txt
python3 tools/fuzz_wide_bpp.py 100 7 > $OUT/fuzz_wide_bpp_100_cases.log 2>&1
for k in 160 192 256 384 512; do python3 tools/wide_run.py 16384 8192 $k BPP 12 1 2>/dev/null | tail -1 >> $OUT/times.txt; done
for k in 192 512; do for it in 4 24; do python3 tools/wide_run.py 16384 8192 $k BPP $it 1 2>/dev/null | tail -1 >> $OUT/times.txt; done; done
python3 tools/wide_run.py 16384 8192 1024 BPP 2 1 2>/dev/null | tail -1 >> $OUT/times.txt
