# kernel table of block pivoting at k = 64 on a mid-size dense matrix (where the passes are short: what is left?)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SMK_BPP_SMALL_ACCURATE=0
for shape in "8192 4096 64" "16384 8192 64" "8192 4096 40"; do
  set -- $shape
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o x -- python3 $ROOT/tools/iter_times.py $1 $2 $3 BPP 40 > $OUT/run_$1_$3.log 2>&1
  DB=$(find $OUT/kt -name '*.db' | head -1)
  python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r04_bpp_$1x$2_k$3_kernel_stats.md > /dev/null
  rm -rf $OUT/kt
  echo "== $shape"; tail -1 $OUT/run_$1_$3.log | cut -c1-200; sed -n 5,12p $OUT/r04_bpp_$1x$2_k$3_kernel_stats.md | cut -c1-150
done
