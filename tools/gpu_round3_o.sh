#!/bin/bash
# round 3, call o: where the time of block principal pivoting above k = 128 goes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03o; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 192 256 512; do
  for alg in; do
    (cd $ROOT && python3 tools/wide_run.py 16384 8192 $k $alg 4 2) 2>/dev/null | tail -1 >> $OUT/times.txt
  done
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$k -o x -- python3 $ROOT/tools/wide_run.py 16384 8192 $k BPP 4 1 > $OUT/run_$k.log 2>&1
  DB=$(find $OUT/kt_$k -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/bpp_k${k}_kernel_stats.md > /dev/null
  rm -rf $OUT/kt_$k
done
