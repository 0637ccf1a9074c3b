#!/bin/bash
# inter-kernel gaps (tools/kernel_gaps.py) of the other workloads
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03l
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
tr() { # name skip count cmd...
  local name=$1 skip=$2 cnt=$3; shift 3
  timeout 600 rocprofv3 --kernel-trace -d $OUT/kt_$name -o x -- "$@" > $OUT/run_$name.log 2>&1
  local DB=$(find $OUT/kt_$name -name '*.db' | head -1)
  echo "=================== $name"
  python3 $ROOT/tools/kernel_gaps.py $DB $skip $cnt | cut -c1-150
  rm -rf $OUT/kt_$name
}
tr c3 600 14 python3 $ROOT/bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline
tr emulate8 400 26 python3 $ROOT/bench.py --emulate-world 8 --steps 10 --warmup 3 --no-cpu-baseline
tr r2_small 2000 16 python3 $ROOT/tools/r2_iter.py 190000 10 300
