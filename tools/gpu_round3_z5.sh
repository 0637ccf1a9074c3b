#!/bin/bash
# round 3, call z5: HALS W sweep by blocks of 16 columns (k > 64) against one full-row launch per column (SMK_HALS_W_BLOCKED=0)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z5; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_variants.py tests/test_sparse.py tests/test_gpu_fullsize.py -m gpu -x -q -k "HALS or hals or above or accurate or fuzz" 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/tests.txt
for t in 0 1 0 1; do
  for k in 100 192 512; do SMK_HALS_W_BLOCKED=$t python3 tools/wide_run.py 16384 8192 $k HALS 12 1 2>/dev/null | tail -1 | sed "s/^/blocked=$t /" >> $OUT/times.txt; done
done
python3 tools/fuzz_parity.py 300 123 2>&1 | tail -1 > $OUT/fuzz.txt
SMK_POISON=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "HALS and above" 2>&1 | grep -E "passed|failed|error" | tail -2 >> $OUT/tests.txt
