#!/bin/bash
# round 3, call z12: block pivoting above k = 64 on the accurate product form -- long runs, parity tests, times
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z12; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
python3 tools/wide_long_run.py 2>&1 | grep -v "amdgpu.ids" > $OUT/long_runs.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_nnls.py tests/test_gpu_fullsize.py -m gpu -x -q -k "BPP or bpp or above or wide or c4 or c2" 2>&1 | grep -E "passed|failed|error" | tail -2 > $OUT/tests.txt
for k in 80 100 128 160 192 256 384 512; do python3 tools/wide_run.py 16384 8192 $k BPP 12 1 2>/dev/null | tail -1 >> $OUT/times.txt; done
python3 tools/wide_run.py 16384 8192 1024 BPP 2 1 2>/dev/null | tail -1 >> $OUT/times.txt
