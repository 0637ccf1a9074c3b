#!/bin/bash
# round 3, call z4: kernel tables at k = 100 (MU, HALS, BPP) on 16384 x 8192
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z4; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for alg in MU HALS BPP; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$alg -o x -- python3 $ROOT/tools/wide_run.py 16384 8192 100 $alg 12 1 > $OUT/run_$alg.log 2>&1
  DB=$(find $OUT/kt_$alg -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/k100_${alg}_kernel_stats.md > /dev/null
  rm -rf $OUT/kt_$alg
done
