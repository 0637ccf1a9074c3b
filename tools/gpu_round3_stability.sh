#!/bin/bash
# round 3: the same block-pivoting runs many times over -- a race shows as an outlier (how the diagonal-block race was found)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03stab; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
for r in 1 2 3 4 5 6 7 8; do
  for k in 100 192 512; do python3 tools/wide_run.py 16384 8192 $k BPP 8 1 2>/dev/null | tail -1 >> $OUT/times.txt; done
  python3 tools/wide_run.py 16384 8192 100 HALS 8 1 2>/dev/null | tail -1 >> $OUT/times.txt
done
