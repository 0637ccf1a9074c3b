#!/bin/bash
# round 3, run b: kernel table of the per-rank work of an 8-rank C4 run (bench.py --emulate-world 8)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in ${CHUNKS:-4}; do
  SMK_COMM_CHUNKS=$c timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_c$c -o x -- python3 $ROOT/bench.py --emulate-world 8 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/run_c$c.log 2>&1
  DB=$(find $OUT/kt_c$c -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r03_c4_emulate8_chunks${c}_kernel_stats.md
  rm -rf $OUT/kt_c$c
done
