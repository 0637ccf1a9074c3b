#!/bin/bash
# package power and shader clock while the k = 64 fp32 streaming kernels run (gpurun_out/r02/r02_k64_f32_power_clock.txt)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02; mkdir -p $OUT
{
echo "rocm-smi --showclocks --showpower sampled every ~0.9 s while tools/mb/mb_bp_sweep 64 8192 262144 0 <variant> loops (MB_REPS=5000)"
echo "variants: 21 = round-1 kernel, 108 = round-2 kernel (bf16x3: 6 products), 120 = the same data path with no MFMA and no bf16 split"
export MB_NSPLIT=3
for v in 21 108 120; do
  (MB_REPS=5000 $ROOT/tools/mb/mb_bp_sweep 64 8192 262144 0 $v > $OUT/long_$v.log 2>&1 &)
  sleep 4
  echo "== variant $v"
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" ; sleep 0.4; done
  while pgrep -x mb_bp_sweep > /dev/null; do sleep 0.5; done
  cat $OUT/long_$v.log; rm -f $OUT/long_$v.log
done
echo "== fp16 two-term form (MB_NSPLIT=4, the default for fp32 A), variant 125 (the default)"
(MB_NSPLIT=4 MB_REPS=5000 $ROOT/tools/mb/mb_bp_sweep 64 8192 262144 0 125 > $OUT/long_f.log 2>&1 &)
sleep 4
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" ; sleep 0.4; done
while pgrep -x mb_bp_sweep > /dev/null; do sleep 0.5; done
cat $OUT/long_f.log; rm -f $OUT/long_f.log
echo "== 2-term fast form (MB_NSPLIT=2), variant 108"
(MB_NSPLIT=2 MB_REPS=5000 $ROOT/tools/mb/mb_bp_sweep 64 8192 262144 0 108 > $OUT/long_f.log 2>&1 &)
sleep 4
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" ; sleep 0.4; done
while pgrep -x mb_bp_sweep > /dev/null; do sleep 0.5; done
cat $OUT/long_f.log; rm -f $OUT/long_f.log
} > $OUT/r02_k64_f32_power_clock.txt 2>&1
cat $OUT/r02_k64_f32_power_clock.txt
