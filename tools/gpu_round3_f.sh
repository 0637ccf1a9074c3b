#!/bin/bash
# C2 (8192 x 4096, k = 16, BPP, fp32): bench line + kernel table
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03f
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_c2.json
python3 -c "
import json; j=json.load(open('$OUT/bench_c2.json')); print('C2 it/s %.1f  us/iter %.1f  bigprod %.1f us'%(j['value'], j['ms_per_step']*1e3, j['roofline']['avg_launch_ms']*1e3))"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o x -- python3 $ROOT/bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/run.log 2>&1
DB=$(find $OUT/kt -name '*.db' | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/prof_summary.py "$DB" $OUT/r03_c2_bpp_f32_kernel_stats.md | head -16 | cut -c1-160
rm -rf $OUT/kt
