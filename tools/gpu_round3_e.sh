#!/bin/bash
# blocked vs unblocked rank-2 gather product on a root-sized matrix: kernel tables
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03e
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for nb in 1 8 4; do
  SMK_SPMM_BLOCKS=$nb timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$nb -o x -- python3 $ROOT/tools/r2_iter.py 1000000 16 30 > $OUT/run_$nb.log 2>&1
  DB=$(find $OUT/kt_$nb -name '*.db' | head -1)
  echo "== SMK_SPMM_BLOCKS=$nb"
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" | head -8 | cut -c1-150
  grep "^rep 1" $OUT/run_$nb.log
  rm -rf $OUT/kt_$nb
done
