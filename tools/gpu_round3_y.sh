#!/bin/bash
# round 3, call y: look-ahead in the tiled Cholesky (workgroup per column) -- parity, fuzz, times; then the bench lines with the
# refreshed traffic file
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03y; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_nnls.py tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_sparse.py -m gpu -x -q -k "above or not_positive or ill_cond or hard or wide or 200 or 150 or 129" 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/tests.txt
python3 tools/fuzz_wide_bpp.py 60 31 2>&1 | tail -2 > $OUT/fuzz.txt
for k in 192 256 384 512; do python3 tools/wide_run.py 16384 8192 $k BPP 12 1 2>/dev/null | tail -1 >> $OUT/times.txt; done
python3 tools/wide_run.py 16384 8192 1024 BPP 2 1 2>/dev/null | tail -1 >> $OUT/times.txt
python3 bench.py 2> $OUT/bench_c4.err | tail -1 > $OUT/r03_bench_c4.json
python3 bench.py --workload c3 --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/r03_bench_c3.json
python3 bench.py --workload c2 --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/r03_bench_c2.json
