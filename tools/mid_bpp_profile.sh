# kernel tables of block pivoting on mid-size dense matrices (where the passes are short: what is left?):
#   bash tools/mid_bpp_profile.sh ["m n k" ...]     default: k = 64 and 40 on 8192 x 4096, k = 64 on 16384 x 8192
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SMK_BPP_SMALL_ACCURATE=0
[ $# -eq 0 ] && set -- "8192 4096 64" "16384 8192 64" "8192 4096 40"
for shape in "$@"; do
  set -- $shape
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o x -- python3 $ROOT/tools/iter_times.py $1 $2 $3 BPP 40 > $OUT/run_$1_$3.log 2>&1
  DB=$(find $OUT/kt -name '*.db' | head -1)
  python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r04_bpp_$1x$2_k$3_kernel_stats.md > /dev/null
  rm -rf $OUT/kt
  echo "== $shape"; grep "steady" $OUT/run_$1_$3.log | cut -c1-140; sed -n 5,13p $OUT/r04_bpp_$1x$2_k$3_kernel_stats.md | cut -c1-150
done
