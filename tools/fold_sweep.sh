# fold interval of the fp16 two-term streaming product (fp32 accumulators added into fp64 every 8 / 4 / 2 / 1 stages):
# product error against an fp64 host product and time per launch (mb_bp_sweep), the C4 bench line per variant, and the
# trajectory distance of tools/long_runs_500.py per variant
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
O=gpurun_out/r04o
if [ "$1" = sweep ]; then
  : > $O/r04_fold_interval_sweep.txt
  for shape in "64 1500 1100" "64 1100 1500" "64 8192 32768" "64 262144 8192" "64 65536 32768" "32 1500 1100"; do
    MB_NSPLIT=4 MB_REPS=20 timeout 300 tools/mb/mb_bp_sweep $shape 0 125 108 128 129 >> $O/r04_fold_interval_sweep.txt 2>&1
  done
fi
if [ "$1" = traj ]; then
  timeout 600 python3 tools/long_runs_500.py "fold every 2" "fold every 1" > $O/r04_fold_interval_trajectories.txt 2>&1
  cat $O/r04_fold_interval_trajectories.txt
fi
if [ "$1" = bench ]; then
  for v in 125 128 108 125 128 129; do
    echo "SMK_BP_VARIANT=$v" >> $O/r04_fold_interval_c4_bench.txt
    SMK_BP_VARIANT=$v timeout 900 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 >> $O/r04_fold_interval_c4_bench.txt
  done
  cat $O/r04_fold_interval_c4_bench.txt
fi
