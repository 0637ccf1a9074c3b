# C2 under rocprofv3 (kernel table) with the pack-in-solve path; args: env assignments to try, e.g. "SMK_TAIL_FIRST=1"
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  export $e
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_c2 -o x -- python3 $ROOT/bench.py --no-cpu-baseline --workload c2 --steps 50 --warmup 5 > $OUT/c2_run.log 2>&1
  DB=$(find $OUT/kt_c2 -name '*.db' | head -1)
  python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r04_c2_bpp_f32_kernel_stats_$e.md > /dev/null
  rm -rf $OUT/kt_c2
  echo "== $e"; sed -n 5,6p $OUT/r04_c2_bpp_f32_kernel_stats_$e.md | cut -c1-160
  for i in 1 2; do python3 $ROOT/bench.py --workload c2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', round(1000*d['ms_per_step'],2), 'us')"; done
  unset ${e%%=*}
done
