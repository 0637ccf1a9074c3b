#!/bin/bash
# round 3, call z3: MU at k in (64, 128] through Y = X G on the matrix cores (A/B: SMK_MU_GEMM128=0)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z3; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_variants.py tests/test_sparse.py -m gpu -x -q -k "MU or mu or fuzz" 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/tests.txt
for t in 0 1 0 1; do
  for k in 80 100 128; do SMK_MU_GEMM128=$t python3 tools/wide_run.py 16384 8192 $k MU 12 1 2>/dev/null | tail -1 | sed "s/^/gemm128=$t /" >> $OUT/times.txt; done
done
python3 tools/fuzz_parity.py 150 99 2>&1 | tail -1 > $OUT/fuzz.txt
