#!/bin/bash
# usage: prof_workload.sh <workload> [steps]   -> kernel-trace stats table for any bench workload
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-c3}; STEPS=${2:-6}
OUT=$ROOT/gpurun_out/prof_$W
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o x -- python3 $ROOT/bench.py --workload $W --steps $STEPS --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/kernel_stats.md | head -20
rm -rf $OUT/kt
