#!/bin/bash
# Throughput while the passive sets move -> gpurun_out/r05/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd $ROOT
python3 tools/active_pivoting.py 262144 65536 64 40 both > $OUT/r05_c4_active_pivoting.txt 2>&1
python3 tools/active_pivoting.py 262144 65536 64 40 both 8 2>&1 | grep -v "^\[" >> $OUT/r05_c4_active_pivoting.txt
python3 tools/active_pivoting.py 8192 4096 16 60 both >> $OUT/r05_c4_active_pivoting.txt 2>&1
cd /tmp
for data in planted uniform; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_piv_$data -o x -- python3 $ROOT/tools/active_pivoting.py 262144 65536 64 30 $data > $OUT/piv_${data}_run.log 2>&1
  DB=$(find $OUT/kt_piv_$data -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r05_c4_${data}_cold_start_kernel_stats.md > /dev/null
  [ -n "$DB" ] && python3 $ROOT/tools/kernel_timeline.py "$DB" nnls_bpp $OUT/r05_c4_${data}_cold_start_nnls_timeline.txt > /dev/null
  rm -rf $OUT/kt_piv_$data
done
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_piv8 -o x -- python3 $ROOT/tools/active_pivoting.py 262144 65536 64 30 planted 8 > $OUT/piv8_run.log 2>&1
DB=$(find $OUT/kt_piv8 -name '*.db' | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r05_c4_planted_rank0_of_8_cold_start_kernel_stats.md > /dev/null
[ -n "$DB" ] && python3 $ROOT/tools/kernel_timeline.py "$DB" nnls_bpp $OUT/r05_c4_planted_rank0_of_8_nnls_timeline.txt > /dev/null
rm -rf $OUT/kt_piv8
