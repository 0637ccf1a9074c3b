#!/bin/bash
# round 3, call z8: Gram partials above k = 128 on 64 x 64 tiles through LDS -- parity, times, kernel table
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z8; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_sparse.py tests/test_gpu_flatclust.py -m gpu -x -q -k "above or wide or 129 or 150 or 200 or 256 or 300 or 700" 2>&1 | grep -E "passed|failed|error" | tail -2 > $OUT/tests.txt
for k in 192 512; do for alg in MU HALS BPP; do python3 tools/wide_run.py 16384 8192 $k $alg 12 1 2>/dev/null | tail -1 >> $OUT/times.txt; done; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o x -- python3 $ROOT/tools/wide_run.py 16384 8192 512 MU 12 1 > $OUT/run.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/mu_k512_kernel_stats.md > /dev/null
rm -rf $OUT/kt
