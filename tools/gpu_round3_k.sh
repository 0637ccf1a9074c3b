#!/bin/bash
# C2: where the time between the kernels goes (inter-kernel gaps from the kernel trace)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03k
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/kt -o x -- python3 $ROOT/bench.py --workload ${1:-c2} --steps 200 --warmup 20 --no-cpu-baseline > $OUT/run.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
python3 $ROOT/tools/kernel_gaps.py $DB 3000 20
rm -rf $OUT/kt
