#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_shard$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd $ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o x -- python3 $ROOT/tools/shard_time.py $1 > $OUT/run.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/kernel_stats.md | head -20
rm -rf $OUT/kt
