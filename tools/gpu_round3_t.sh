#!/bin/bash
# round 3, call t: after the fix of the diagonal-block race in the panel Cholesky -- tests, then k >= 512 three times over
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03t; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_nnls.py tests/test_gpu_parity.py -m gpu -x -q -k "above or not_positive or ill_cond or hard" 2>&1 | tail -3 > $OUT/tests.txt
for round in 1 2 3; do
  for k in 192 512; do python3 tools/wide_run.py 16384 8192 $k BPP 12 1 2>/dev/null | tail -1 >> $OUT/times.txt; done
  python3 tools/wide_run.py 16384 8192 1024 BPP 2 1 2>/dev/null | tail -1 >> $OUT/times.txt
done
python3 tools/fuzz_wide_bpp.py 60 11 > $OUT/fuzz_wide_bpp_60_cases.log 2>&1
