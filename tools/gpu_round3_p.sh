#!/bin/bash
# round 3, call p: block principal pivoting above k = 128, first iterations against later ones
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03p; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
for k in 192 256 512; do
  for it in 4 12 24; do
    python3 tools/wide_run.py 16384 8192 $k BPP $it 1 2>/dev/null | tail -1 >> $OUT/times.txt
  done
done
