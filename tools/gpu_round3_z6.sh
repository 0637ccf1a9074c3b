#!/bin/bash
# round 3, call z6: kernel tables of HALS at k = 100 and k = 512 with the blocked W sweep
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z6; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 100 512; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$k -o x -- python3 $ROOT/tools/wide_run.py 16384 8192 $k HALS 12 1 > $OUT/run_$k.log 2>&1
  DB=$(find $OUT/kt_$k -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/hals_k${k}_kernel_stats.md > /dev/null
  [ -n "$DB" ] && python3 $ROOT/tools/kernel_gaps.py "$DB" > $OUT/hals_k${k}_gaps.txt 2>&1
  rm -rf $OUT/kt_$k
done
