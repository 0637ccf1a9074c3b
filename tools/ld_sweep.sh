# Does a leading dimension that is a large power of two cost bandwidth?  The streaming product on len and len + 128 rows
# (512 B / 256 B more per column), fp32 A in the fp16 two-term form and bf16 A, two repetitions each.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04r
O=gpurun_out/r04r/r04_leading_dimension_sweep.txt
: > $O
for rep in 1 2; do
for len in 16384 32768 65536 131072 262144; do
  nc=$((2147483648 / len / 2)); [ $nc -gt 65536 ] && nc=65536
  for l in $len $((len + 128)); do
    MB_NSPLIT=4 MB_REPS=10 timeout 300 tools/mb/mb_bp_sweep 64 $l $nc 0 125 2>&1 | cut -c1-110 >> $O
  done
done
for len in 16384 65536; do
  nc=$((1073741824 / len)); [ $nc -gt 65536 ] && nc=65536
  for l in $len $((len + 128)); do
    MB_NSPLIT=3 MB_REPS=10 timeout 300 tools/mb/mb_bp_sweep 32 $l $nc 1 6 2>&1 | cut -c1-110 >> $O
  done
done
done
cat $O
