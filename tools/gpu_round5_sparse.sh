#!/bin/bash
# Sparse NMF at k > 2: bench lines and kernel tables of the s_* workloads -> gpurun_out/r05/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
kt() {   # name, command...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o x -- "$@" > $OUT/${name}_run.log 2>&1
  local DB=$(find $OUT/kt_$name -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r05_${name}_kernel_stats.md > /dev/null
  rm -rf $OUT/kt_$name
}
cd $ROOT
for w in s_reuters s_reuters_hals s_1m; do
  python3 $ROOT/bench.py --workload $w --steps 50 --warmup 10 2> $OUT/bench_$w.err | tail -1 > $OUT/r05_bench_$w.json
  SMK_SPMM_SEG=0 python3 $ROOT/bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r05_bench_${w}_round4_kernel.json
done
for w in s_reuters s_reuters_hals s_1m; do
  kt $w python3 $ROOT/bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline
done
