#!/bin/bash
# Cache policy of the streamed loads of A by matrix size (SMK_BP_TEMPORAL=0: non-temporal, 1: default policy): steady-state
# iteration time of BPP k = 16 on fp32 matrices from 67 MB to 1 GB per copy, and HALS k = 32 bf16 -> gpurun_out/r05/r05_cache_policy_ab.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
cd $ROOT
{
echo "# steady-state ms per iteration (tools/active_pivoting.py, iterate(1) + sync), A + A' streamed per iteration in MB"
for shape in "8192 2048" "8192 3072" "8192 4096" "8192 5120" "8192 6144" "8192 8192" "16384 8192" "32768 8192"; do
  set -- $shape
  mb=$(( $1 * $2 * 4 * 2 / 1000000 ))
  for t in 0 1; do
    echo -n "fp32 $1 x $2 BPP k=16 ($mb MB) temporal=$t: "
    SMK_BP_TEMPORAL=$t python3 tools/active_pivoting.py $1 $2 16 60 uniform 2>/dev/null | head -1 | sed 's/.*steady state/steady state/'
  done
done
} > $OUT/r05_cache_policy_ab.txt 2>&1
