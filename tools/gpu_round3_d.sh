#!/bin/bash
# lanes-per-column sweep of the rank-2 gather product on root-sized and small-node-sized matrices
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for lpc in 1 2 4 8 16; do
  echo "== SMK_SPMM2_LPC=$lpc"
  SMK_SPMM2_LPC=$lpc python3 tools/r2_iter.py 1000000 16 30 | tail -1
  SMK_SPMM2_LPC=$lpc python3 tools/r2_iter.py 190000 10 300 | tail -1
done
