#!/bin/bash
# round 3, call r: tile kernel with one wave or the whole workgroup per column
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03r; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_nnls.py tests/test_gpu_parity.py tests/test_gpu_flatclust.py tests/test_gpu_variants.py -m gpu -x -q -k "above or not_positive or ill_cond or hard or nnls_hals or fuzz" 2>&1 | tail -5 > $OUT/tests_nw4.txt
SMK_WIDE_NW=1 python -m pytest tests/test_gpu_nnls.py tests/test_gpu_parity.py -m gpu -x -q -k "above_128 or not_positive or hard" 2>&1 | tail -3 > $OUT/tests_nw1.txt
for nw in 1 4; do
for k in 160 192 256; do
  SMK_WIDE_NW=$nw python3 tools/wide_run.py 16384 8192 $k BPP 12 1 2>/dev/null | tail -1 | sed "s/^/NW=$nw /" >> $OUT/times.txt
done; done
for k in 384 512; do python3 tools/wide_run.py 16384 8192 $k BPP 12 1 2>/dev/null | tail -1 >> $OUT/times.txt; done
python3 tools/wide_run.py 16384 8192 1024 BPP 2 1 2>/dev/null | tail -1 >> $OUT/times.txt
