#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_c5
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd $ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o x -- python3 $ROOT/tools/c5_hier.py 1000000 16 8 > $OUT/run.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/kernel_stats.md | head -24
tail -3 $OUT/run.log
rm -rf $OUT/kt
