#!/bin/bash
# round 3, call q: the panel Cholesky of the Gram matrix (ranks above 128) -- tests, then timing against the one-workgroup kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03q; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_nnls.py tests/test_gpu_parity.py tests/test_gpu_flatclust.py -m gpu -x -q -k "above or not_positive or ill_cond or hard or nnls_hals" 2>&1 | tail -5 > $OUT/tests.txt
for k in 192 256 512 1024; do
  it=12; [ $k = 1024 ] && it=2
  python3 tools/wide_run.py 16384 8192 $k BPP $it 1 2>/dev/null | tail -1 >> $OUT/times.txt
done
cd /tmp && export TMPDIR=/tmp
for k in 192 512; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$k -o x -- python3 $ROOT/tools/wide_run.py 16384 8192 $k BPP 4 1 > $OUT/run_$k.log 2>&1
  DB=$(find $OUT/kt_$k -name '*.db' | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/bpp_k${k}_kernel_stats.md > /dev/null
  rm -rf $OUT/kt_$k
done
