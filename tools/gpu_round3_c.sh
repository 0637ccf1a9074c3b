#!/bin/bash
# round 3, run c: kernel tables of the RANK2 iteration on C5-shaped sparse matrices (root-sized and small-node-sized)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03c
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for cfg in "1000000 16 30" "190000 10 300"; do
  set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$1 -o x -- python3 $ROOT/tools/r2_iter.py $1 $2 $3 > $OUT/run_$1.log 2>&1
  DB=$(find $OUT/kt_$1 -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r2_iter_$1_kernel_stats.md | head -16
  grep "^rep" $OUT/run_$1.log
  rm -rf $OUT/kt_$1
done
