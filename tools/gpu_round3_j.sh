#!/bin/bash
# blocked rank-2 gather product: segments per lane group in flight (SMK_SPMM_UNROLL) x lanes per segment, root-sized matrix
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for u in 1 2 4; do for lpc in 1 2 4; do
  echo -n "unroll $u lanes $lpc: "; SMK_SPMM_UNROLL=$u SMK_SPMM_BLOCKED_LPC=$lpc python3 tools/r2_iter.py 1000000 16 30 | tail -1
done; done
