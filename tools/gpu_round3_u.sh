#!/bin/bash
# round 3, call u: C2 with reduce + pack as one launch (A/B), parity
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03u; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or iteration or c2" 2>&1 | tail -3 > $OUT/tests.txt
for rp in 1 0 1 0; do
  SMK_REDUCE_PACK=$rp python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('reduce_pack=$rp', d['value'], 'it/s', d['ms_per_step']*1000, 'us per iteration')" >> $OUT/c2.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o x -- python3 $ROOT/bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/run.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/c2_kernel_stats.md > /dev/null
rm -rf $OUT/kt
