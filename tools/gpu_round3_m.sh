#!/bin/bash
# inter-kernel gaps of the C5-shaped HierNMF2 run (small-node iterations with the stopping rule)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03m
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o x -- python3 $ROOT/tools/c5_hier.py 1000000 16 8 > $OUT/run.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
python3 $ROOT/tools/kernel_gaps.py $DB 60000 16 | cut -c1-150
rm -rf $OUT/kt
