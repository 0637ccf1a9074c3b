#!/bin/bash
# round 3: rank sweep and kernel tables above k = 64 on the last build (block pivoting with accurate products)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03ev; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
for k in 80 100 128 160 192 256 384 512; do for alg in BPP MU HALS; do python3 tools/wide_run.py 16384 8192 $k $alg 12 1 2>/dev/null | tail -1; done; done > $OUT/r03_wide_rank_times.txt
python3 tools/wide_run.py 16384 8192 1024 BPP 2 1 2>/dev/null | tail -1 >> $OUT/r03_wide_rank_times.txt
cd /tmp && export TMPDIR=/tmp
for k in 100 192 512; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$k -o x -- python3 $ROOT/tools/wide_run.py 16384 8192 $k BPP 12 1 > $OUT/run_$k.log 2>&1
  DB=$(find $OUT/kt_$k -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r03_wide_bpp_k${k}_kernel_stats.md > /dev/null
  rm -rf $OUT/kt_$k $OUT/run_$k.log
done
