#!/bin/bash
# round 3, call x: the accurate-form product with its tile loop unrolled (four independent accumulator chains) -- parity, times
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03x; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_flatclust.py tests/test_gpu_variants.py -m gpu -x -q -k "accurate or above or HALS or hals or dynamic or magnitude" 2>&1 | tail -4 > $OUT/tests.txt
for k in 100 192 512; do python3 tools/wide_run.py 16384 8192 $k HALS 12 1 2>/dev/null | tail -1 >> $OUT/times.txt; done
SMK_NSPLIT=8 python3 bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 with the accurate form:', d['value'], 'it/s', d['roofline']['avg_launch_ms'], 'ms per pass')" >> $OUT/times.txt
SMK_NSPLIT=8 python3 bench.py --workload c4s --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 shard with the accurate form:', d['value'], 'it/s', d['roofline']['avg_launch_ms'], 'ms per pass')" >> $OUT/times.txt
