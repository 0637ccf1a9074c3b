#!/bin/bash
# segment length / unroll sweep of the sparse gather product (tools/spmm_rate.py) -> gpurun_out/r05/spmm_sweep.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
cd $ROOT
: > $OUT/spmm_sweep.txt
for seg in 16 32 64 128; do for u in 4 8 16; do
  SMK_SPMM_SEG_LEN=$seg SMK_SPMM_SEG_U=$u python3 tools/spmm_rate.py ${1:-both} 32 2>/dev/null >> $OUT/spmm_sweep.txt
done; done
SMK_SPMM_SEG=0 python3 tools/spmm_rate.py ${1:-both} 32 2>/dev/null >> $OUT/spmm_sweep.txt
